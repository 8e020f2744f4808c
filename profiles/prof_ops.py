import sys, torch
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--workload", "3d", "--steps", "2", "--warmup", "2", "--no-cpu-baseline"]
from torch.profiler import profile, ProfilerActivity
import runpy
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=False) as prof:
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
