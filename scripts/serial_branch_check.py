#!/usr/bin/env python3
"""bench.py with the sequence branch serialised on the main stream (no side stream): extra arguments are passed on.
Debug aid for stream-ordering problems: `python scripts/serial_branch_check.py --no-cpu-baseline [--no-graph]`."""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import analysisgnn_amd.encoders as enc
enc._HybridMixin.overlap_sequence_branch = False
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
