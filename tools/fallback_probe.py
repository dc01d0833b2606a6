#!/usr/bin/env python3
"""The sharded bench with the library's RCCL communicator made to fail on every rank: every rank must fall back to the
Python-orchestrated driver together and still print the line.  Run under torch.distributed.run (tools/gpu.sh run:...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alaz_amd import engine

class NoRccl:
    def __init__(self, *a, **k): raise engine.ServiceGraphError(engine.SG_ENODEV, "RCCL made unavailable by tools/fallback_probe.py")
engine.RcclComm = NoRccl
os.environ["SG_FORCE_SHARDED"] = "1"
sys.argv = [sys.argv[0], "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"]
import bench
sys.exit(bench.main() or 0)
