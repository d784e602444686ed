"""CPU oracle (test infrastructure only -- see trackmpnn_oracle.py)."""
