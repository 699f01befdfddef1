"""CPU oracle of the cumf_als solve path -- test infrastructure only (see als_oracle.c)."""
