"""Time the training step (configs[1]) under lab switches that bench.py does not offer (timing only where noted):
    python tools/lab/step_time.py [--lib tools/lab/libcrct_lab.so] [--steps N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
    sys.path.insert(0, p)
import torch                                             # noqa: E402
from crct import config as CFG, synthetic as S           # noqa: E402
from crct.model import VisualDialogEncoder               # noqa: E402
from crct.optim import get_optimizer, WarmupLinearScheduleNonZero, FusedAdamW   # noqa: E402
from crct.step_adapter import forward as step_forward    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="", help="another build of libcrct_hip.so (A/B builds under tools/lab/)")
    ap.add_argument("--bf16-grads", action="store_true", help="timing only: Linear weight gradients written as bf16 (CrctStepCfg.grads_bf16) and AdamW "
                    "reading the bf16 buffer for EVERY element (the small fp32-accumulated gradients are not in it: wrong numerics)")
    ap.add_argument("--emb-late", type=int, default=0, help="timing only: all but the last N blocks of the embedding segment's AdamW launch run at the "
                    "END of the update sequence (the forward then waits for N blocks only and may read rows that are not updated yet)")
    ap.add_argument("--residual-bf16", action="store_true", help="the residual stream stored as bf16 (rounds 1 - 5; params['residual_fp32'] = False)")
    ap.add_argument("--no-word-index", action="store_true", help="the word-embedding gradient by the scanning kernel (crct_embed_word_index(0)): same bits, round-5 speed")
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--vis", type=int, default=36)
    ap.add_argument("--tokens", type=int, default=20)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    if a.lib:
        from crct import lib as L
        L.LIB_PATH = os.path.abspath(a.lib)
    dev = torch.device("cuda", 0)
    cfg = CFG.vilbert_config(v_feature_size=2048)
    params = CFG.default_params(device=dev, batch_size=a.batch, seed=0)
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.sync_stats = False
    if a.no_word_index:
        from crct import lib as L2
        L2.load().crct_embed_word_index(0)
    core.residual_fp32 = not a.residual_bf16
    core.stream_mode = (1, 1)
    model.train()
    opt = get_optimizer(params, model)
    opt.overlap = True
    if a.bf16_grads:
        buf = torch.zeros(core.flat_grads.numel(), dtype=torch.bfloat16, device=dev)
        from crct.engine import StepEngine
        eng_backward = StepEngine.backward

        def backward(self, p32, p16, g32, tensors, step, seg):
            if step.get("wgrad_overwrite"):
                step = dict(step, grads_bf16=buf)
            return eng_backward(self, p32, p16, g32, tensors, step, seg)
        StepEngine.backward = backward
        launch = FusedAdamW._launch

        def _launch(self, b0, b1, inv_scale, stream, max_workgroups=0):
            self._g16 = buf
            return launch(self, b0, b1, inv_scale, stream, max_workgroups)
        FusedAdamW._launch = _launch
    if a.emb_late:
        launch0 = FusedAdamW._launch
        stash = {}

        def _launch2(self, b0, b1, inv_scale, stream, max_workgroups=0):
            n = len(self._seg_blocks)
            if (b0, b1) == tuple(self._seg_blocks[n - 1]) and b1 - b0 > a.emb_late:
                stash["r"] = (b0, b1 - a.emb_late)
                return launch0(self, b1 - a.emb_late, b1, inv_scale, stream, max_workgroups)
            r = launch0(self, b0, b1, inv_scale, stream, max_workgroups)
            if (b0, b1) == tuple(self._seg_blocks[0]) and "r" in stash:
                c0, c1 = stash.pop("r")
                launch0(self, c0, c1, inv_scale, stream, self.overlap_workgroups)
            return r
        FusedAdamW._launch = _launch2
    sched = WarmupLinearScheduleNonZero(opt, warmup_steps=params["warmup"], t_total=60000, min_lr=params["min_lr"])
    pool = [{k: v.to(dev) for k, v in S.make_batch(a.batch, a.tokens, a.vis, 2048, seed=1234 + 97 * i).items()} for i in range(8)]

    def step(i):
        loss = step_forward(model, pool[i % 8], params)[0]
        loss.backward()
        opt.step()
        opt.zero_grad()
        sched.step()
    for i in range(8):
        step(i)
    out = []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / a.steps * 1e3)
    print("step_time emb_late=%d bf16_grads=%s residual=%s word_index=%d lib=%s: %s ms" % (a.emb_late, a.bf16_grads, "bf16" if a.residual_bf16 else "fp32", not a.no_word_index, os.path.basename(a.lib) or "product", ", ".join("%.3f" % x for x in out)), flush=True)


if __name__ == "__main__":
    main()
