#!/usr/bin/env python3
"""`python main4real.py -i datasets/<name> ...` -- flags of the reference's main4real.py:12-32 (no ground truth)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dual_dmp_amd.cli import main4real  # noqa: E402

if __name__ == "__main__":
    main4real()
