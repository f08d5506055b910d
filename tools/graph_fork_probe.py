"""Does hipGraph capture take the data-parallel step's stream program - three streams (main, side, comm), the comm stream joined by
events several times - in ONE capture?  (train_step.ESRGANTrainStep._dp_step_body, SSR_DP_ONE_GRAPH=1).  Variants by argv[1]:
  full      the pattern of _dp_step_body
  nocomm    collectives' stand-ins on the issuing stream (no comm stream)
  onejoin   comm stream used, but every consumer waits with wait_stream (no cross-stream events)
  curonly   the comm stream is forked from the main stream only (the side chain runs its exchange in line); curonly1: one fork"""
import faulthandler, sys
import torch
faulthandler.enable()
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
x = [torch.zeros(1 << 16, device="cuda") for _ in range(8)]
cur = torch.cuda.current_stream()
side, cs = torch.cuda.Stream(), torch.cuda.Stream()


def exchange(t, inline=False):
    if mode == "nocomm" or inline:
        t.add_(1.0)
        return None
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        t.add_(1.0)
        ev = torch.cuda.Event()
        ev.record(cs)
    return ev


def wait(ev):
    if ev is None:
        return
    if mode == "onejoin":
        torch.cuda.current_stream().wait_stream(cs)
    else:
        torch.cuda.current_stream().wait_event(ev)


def body():
    x[0].mul_(2.0)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        x[1].add_(x[0])
    hs = []
    for k in range(1 if mode == "curonly1" else 3):
        x[2 + k].add_(x[0])
        hs.append(exchange(x[2 + k]))
    with torch.cuda.stream(side):
        h = exchange(x[1], inline=mode in ("curonly", "curonly1"))
        wait(h)
        x[5].add_(x[1])
    for h in hs:
        wait(h)
    x[6].add_(x[2] + x[3] + x[4])
    cur.wait_stream(side)
    if mode != "nocomm":
        cur.wait_stream(cs)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    cur = torch.cuda.current_stream()
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        body()
    g.replay()
    torch.cuda.synchronize()
print(mode, "captured and replayed:", [float(t[0]) for t in x[:7]])
