"""Command-line front end of micromix_amd/_check_acc_regs.py (the accumulator-register guard of the tile kernels):
python tools/check_acc_regs.py [-DFLAG ...]  -- exit code 1 on a violation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micromix_amd._check_acc_regs import *   # noqa: F401,F403  (tests and tools import check / check_counted / verify from here)
from micromix_amd._check_acc_regs import main

if __name__ == "__main__":
    sys.exit(main())
