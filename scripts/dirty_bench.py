"""bench.py after 24 GiB of device memory were filled with a pattern that is a wild index in every work list and
returned to the driver: the warm-up (work lists growing through overflowing evaluations) must not depend on what
freshly allocated memory holds.  usage (GPU box, repo root): python scripts/dirty_bench.py"""
import sys, torch
# dirty 24 GiB of device memory with a pattern that is a wild index in every work list, then give it back to the driver
xs = [torch.full((2**30,), 0x7f7f7f7f, dtype=torch.int32, device="cuda") for _ in range(6)]
torch.cuda.synchronize()
del xs
torch.cuda.empty_cache()
import runpy
sys.argv = ["bench.py", "--cpu-seconds", "0", "--steps", "30"]
runpy.run_path("bench.py", run_name="__main__")
