"""Developer probe: does hipGraph capture / replay of the engine's forward and backward work in a given stream mode?
Each mode runs in a child process (a crash inside the HIP runtime must not take the probe down):

    python tools/graph_probe.py            # parent: spawns the children, never touches the GPU itself
    python tools/graph_probe.py child <streams> <overlap>
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(streams, overlap):
    os.environ["CRCT_STREAMS"] = streams
    for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
        sys.path.insert(0, p)
    import torch
    from crct import config as CFG, synthetic as S, lib as L
    from crct.model import VisualDialogEncoder
    from crct.optim import get_optimizer
    from crct.step_adapter import forward as step_forward
    import ctypes as C

    dev = torch.device("cuda:0")
    cfg = CFG.vilbert_config(v_feature_size=2048)
    params = CFG.default_params(device=dev, batch_size=80, seed=0)
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.sync_stats = False
    model.train()
    opt = get_optimizer(params, model)
    opt.overlap = bool(int(overlap))
    pool = [{k: v.to(dev) for k, v in S.make_batch(80, 20, 36, 2048, seed=1234 + i).items()} for i in range(4)]
    it = [0]

    def run_step():
        b = pool[it[0] % len(pool)]
        it[0] += 1
        loss = step_forward(model, b, params)[0]
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = run_step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, float(loss)

    for _ in range(3):
        run_step()
    ms, loss = timed(10)
    print("[probe] streams=%s overlap=%s eager: %.3f ms/step loss %.6f" % (streams, overlap, ms, loss), flush=True)
    core.use_graph = True
    for i in range(4):
        ms1, loss = timed(1)
        print("[probe]   graph step %d: %.3f ms loss %.6f" % (i, ms1, loss), flush=True)
    ms, loss = timed(10)
    k, x, br = C.c_int(), C.c_int(), C.c_int()
    L.load().crct_engine_graph_stats(core._engine.handle, C.byref(k), C.byref(x), C.byref(br))
    print("[probe] streams=%s overlap=%s graph: %.3f ms/step loss %.6f keys=%d instantiated=%d broken=%d err=%r"
          % (streams, overlap, ms, loss, k.value, x.value, br.value, L.load().crct_last_error().decode()), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], sys.argv[3])
        return
    modes = sys.argv[1:] or ["0:0", "1:0", "2:0", "2:1"]
    for m in modes:
        s, o = m.split(":")
        env = dict(os.environ, CRCT_DEBUG="1")
        r = subprocess.run(["timeout", "300", sys.executable, os.path.abspath(__file__), "child", s, o], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        tail = "\n".join(l for l in r.stdout.splitlines() if "[probe]" in l or "[crct]" in l or "rror" in l)[-6000:]
        print("==== mode streams=%s overlap=%s rc=%d\n%s" % (s, o, r.returncode, tail), flush=True)


if __name__ == "__main__":
    main()
