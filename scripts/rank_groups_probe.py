import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from permafrost_engine_amd import tick
for world, rank in ((4,0),(4,1),(4,2),(4,3),(2,0),(8,3)):
    T = tick.NavTick(rank=rank, world=world, shared_map=True, fields_per_rank=64//world, agents_per_rank=100000//world, pipeline_fields=True)
    T._comm_pending=False; T.pipelined=False
    for _ in range(4): T.compute()
    T.sync()
    g = bench.profiled_ticks.__wrapped__(T, 4) if hasattr(bench.profiled_ticks,'__wrapped__') else None
    import time
    t0=time.perf_counter()
    for _ in range(30): T.compute()
    T.sync(); dt=(time.perf_counter()-t0)/30
    T.ctx.set_profiling(True)
    rows=[]
    keep=T.pipeline_fields; T.pipeline_fields=False; T.overlap=False
    for _ in range(4):
        T.compute(); T.sync(); rows.append(T.ctx.last_step_ms())
    T.ctx.set_profiling(False)
    print(world, rank, "ms/tick %.4f"%(dt*1e3), "groups", [round(x,3) for x in rows[-1]], "lists", T.ctx.last_step_lists(), flush=True)
    T.close()
