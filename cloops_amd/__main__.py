"""`python -m cloops_amd ...` (see cloops_amd.pipe.main)."""
import os
import sys

# the sweep keeps one stream per chromosome busy: three hardware queues serve them best; an
# explicit setting of the user wins.  Must be in the environment before the first HIP call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")

from .pipe import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
