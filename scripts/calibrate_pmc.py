"""Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: known byte counts in the two access patterns."""
import sys
import shasta_amd
lib = shasta_amd.load()
for nbytes in (1 << 30, 3 << 30):
    lib.calibrate(nbytes, 0)
    lib.calibrate(nbytes, 1)
print("calibration kernels done")
