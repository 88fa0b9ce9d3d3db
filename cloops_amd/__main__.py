"""`python -m cloops_amd ...` (see cloops_amd.pipe.main)."""
import sys

from .pipe import main

if __name__ == "__main__":
    sys.exit(main())
