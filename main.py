#!/usr/bin/env python3
"""`python main.py -i datasets/<name> ...` -- flags of the reference's main.py:15-37 (see dual-dmp_amd/cli.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dual_dmp_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
