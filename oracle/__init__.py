"""CPU oracle for the MI-Seg hot path (test infrastructure; see oracle/functional.py header)."""
