"""CPU oracle package -- test infrastructure only (see qs_oracle.py)."""
