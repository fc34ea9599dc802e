"""What does handing a batch to the captured step cost?  (GPU box)
Times the replayed training step of a workload with
  copy    the staged batch copied into the plan's input arena in front of every step (bench.py)
  none    no copy (the batch of the step before stays in the arena): the floor
  side    the copy issued on a second stream into ANOTHER buffer while the step before runs,
          the step's stream waits for its event (what a double-buffered hand-over would pay
          at the step boundary; the in-graph copy into the arena is not included)
usage: python tools/input_gap.py [workload] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from elektronn2_amd import nets

wl = sys.argv[1] if len(sys.argv) > 1 else "lite183"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
builder, sp, _ = bench.WORKLOADS[wl]
np.random.seed(1)
model = getattr(nets, builder)((None, 1) + sp)
osp = tuple(model.prediction_node.shape.spatial_shape)
model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
opt = model.optimisers['Adam']
opt.step.compile()
plan = opt.step.func
rng = np.random.RandomState(0)
x = torch.tensor(rng.rand(1, 1, *sp).astype(np.float32), device="cuda")
t = torch.tensor(rng.randint(0, 2, (1, 1) + osp).astype(np.float32), device="cuda")
plan.set_inputs([x, t])
arena = plan.input_arena
staged = [arena.clone() for _ in range(4)]
other = torch.empty_like(arena)
side = torch.cuda.Stream()
ev = torch.cuda.Event()


def one_step(i, mode):
    if mode == "side":
        side.wait_stream(plan.stream) if i == 0 else None
        with torch.cuda.stream(side):
            other.copy_(staged[i % 4], non_blocking=True)
            ev.record(side)
        plan.stream.wait_event(ev)
    with torch.cuda.stream(plan.stream):
        if mode == "copy":
            arena.copy_(staged[i % 4], non_blocking=True)
        opt._ensure_state(plan)
        opt._sync_hyper(plan)
    plan.run()


for i in range(5):
    one_step(i, "copy")
torch.cuda.synchronize()
ctx = plan.ctx
for rep in range(2):
    for mode in ("copy", "none", "side"):
        e0, e1 = ctx.event(), ctx.event()
        old = ctx.stream
        torch.cuda.synchronize()
        ctx.set_stream(plan.stream); ctx.record(e0); ctx.set_stream(old)
        for i in range(steps):
            one_step(i, mode)
        ctx.set_stream(plan.stream); ctx.record(e1); ctx.set_stream(old)
        torch.cuda.synchronize()
        print("%s %-5s %.4f ms per step" % (wl, mode, ctx.elapsed_ms(e0, e1) / steps), flush=True)
