"""Stand-in worker for tests/test_bench_spawn_cpu.py: behaves like a bench.py rank as far as the parent can see."""
import json
import os
import sys
import time

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
if mode == "fail-rank1":
    if rank == 1:
        sys.stderr.write("rank 1: no such device\n")
        sys.exit(3)
    time.sleep(120)  # rank 0 would sit in the rendezvous; the parent has to end it
if mode == "stuck-side-leg":
    # what bench.py does when the sharded videocompare leg hangs in its collective: rank 0 prints the headline with an error entry
    # and leaves with EXIT_SIDE_LEG_STUCK; a rank that is hung for good never leaves by itself
    if rank == 0:
        print(json.dumps({"n_gpus": world, "config": {"error": "side leg stuck"}}), flush=True)
        sys.exit(5)
    time.sleep(120)
if rank == 0:
    print(json.dumps({"n_gpus": world, "local_rank": os.environ["LOCAL_RANK"], "master": os.environ["MASTER_ADDR"]}), flush=True)
