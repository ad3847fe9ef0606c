"""Test infrastructure (not product): runs bench.py's main() with the oracle's CPU operator restatement installed in place of the HIP
operator backend, so the launcher / rank wiring / gloo gradient reduction / JSON contract of `python bench.py --gpus N` can be exercised
end to end on a box without a GPU (tests/test_bench_launch.py).  bench.py re-execs `sys.argv[0]` under torch.distributed.run, so every
rank of the child job comes back through this file."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MANTIS_BENCH_DEVICE"] = "cpu"

import mantis_amd.engine as eng      # noqa: E402
import mantis_amd.optim as opt       # noqa: E402
from oracle import ops_ref           # noqa: E402

eng.K = ops_ref
opt.K = ops_ref

import bench                         # noqa: E402

if __name__ == "__main__":
    bench.main()
