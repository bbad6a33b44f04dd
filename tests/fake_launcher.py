"""Stand-in for torch.distributed.run in tests/test_bench_launch.py: 'fails' (or stalls) unless the
command line asks for the torch collective, then prints one JSON line like rank 0 would."""
import json
import os
import sys
import time

if "--collective" in sys.argv and sys.argv[sys.argv.index("--collective") + 1] == "torch":
    print(json.dumps({"metric": "stub", "config": {"collective": "torch"}, "argv": sys.argv[1:]}))
    sys.exit(0)
if os.environ.get("FAKE_LAUNCHER_STALL"):
    time.sleep(60)
sys.exit(3)
