"""Parity oracle package (test infrastructure only; see oracle/csr_oracle.c)."""
