import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, rovmpc
from rovmpc.closed_loop import run_closed_loop
eng = rovmpc.Engine(rovmpc.MPCConfig(N=20, K=4096))
def t(tag):
    run_closed_loop(eng, 12, 200, feedback=True, mode="pipelined")
    r = run_closed_loop(eng, 12, 3000, feedback=True, mode="pipelined")
    print(tag, round(1e6 * r.wall_s / r.steps, 2), flush=True)
which = sys.argv[1]
dev = torch.device("cuda", 0)
state, U = rovmpc.synthetic_problem(4096, 20)
d_state = torch.tensor(state, device=dev); d_U = torch.tensor(U, device=dev)
if "pre" in which:
    t("pre")
if which != "none":
    cfg2 = rovmpc.MPCConfig(N=20, K=4096, threads_per_block=256 if "nt256" in which else 0)
    enga = rovmpc.Engine(cfg2); eng2 = rovmpc.Engine(cfg2)
    st2 = torch.cuda.Stream(device=dev, priority=-1)
    stream = torch.cuda.current_stream()
    pair = [(enga, stream), (eng2, st2)]
    r2 = torch.empty((4, enga.result_len), dtype=torch.float64, device=dev)
    for i in range(600):
        e, st = pair[i & 1]
        e.step_device(d_state.data_ptr(), d_U.data_ptr(), r2[i & 3].data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    if "close" in which:
        eng2.close(); enga.close()
t("after " + which)
