import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, rovmpc
from rovmpc.closed_loop import run_closed_loop
eng = rovmpc.Engine(rovmpc.MPCConfig(N=20, K=4096))
def t(tag):
    run_closed_loop(eng, 12, 200, feedback=True, mode="pipelined")
    r = run_closed_loop(eng, 12, 3000, feedback=True, mode="pipelined")
    print(tag, round(1e6 * r.wall_s / r.steps, 2), flush=True)
which = sys.argv[1]
x = torch.zeros(1024, device="cuda")
if which.endswith("_first"):
    s = torch.cuda.Stream(priority=-1 if which.startswith("hi") else 0)
    with torch.cuda.stream(s):
        x += 1
    torch.cuda.synchronize()
    if which.startswith("hi2"):
        s2 = torch.cuda.Stream(priority=-1)
        with torch.cuda.stream(s2):
            x += 1
        torch.cuda.synchronize()
t("baseline")
if which == "hi_idle":
    s = torch.cuda.Stream(priority=-1)
elif which == "hi_used":
    s = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(s):
        x += 1
    torch.cuda.synchronize()
elif which == "lo_used":
    s = torch.cuda.Stream(priority=0)
    with torch.cuda.stream(s):
        x += 1
    torch.cuda.synchronize()
elif which == "engines":
    e2 = rovmpc.Engine(rovmpc.MPCConfig(N=20, K=4096)); e3 = rovmpc.Engine(rovmpc.MPCConfig(N=20, K=4096)); e2.close(); e3.close()
t("after " + which)
t("again")
