"""Worker of tests/test_gpu_rccl.py: ONE rank on cuda:0 with the real `nccl` (= RCCL) backend.

A one-rank process group exercises everything of the data-parallel path that does not need a second device:
`init_process_group("nccl", device_id=...)`, the flat gradient all-reduces of FlatAdam (the 3D one asynchronously on a side
stream, as bench.py issues it), and synchronised BatchNorm's all_gather / all_reduce -- whose results with one rank must equal
rank-local BatchNorm.  Exit code 0 = all checks passed (details on stderr otherwise).  MOPA_FORCE_COLLECTIVES=1 is required
(a one-rank group would otherwise skip the collectives).
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mopa_amd import syncbn, synth  # noqa: E402
from mopa_amd.config import default_cfg  # noqa: E402
from mopa_amd.models.build import build_model_2d, build_model_3d  # noqa: E402
from mopa_amd.optim import FlatAdam  # noqa: E402
from oracle.params import det_tensor  # noqa: E402  (test infrastructure: deterministic weights keyed by name)


def main():
    assert os.environ.get("MOPA_FORCE_COLLECTIVES") == "1"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    failures = []
    cfg = default_cfg(5, True)
    b = synth.collate([synth.make_scan(31, H=64, W=96), synth.make_scan(32, H=64, W=96)])
    locs, feats = b["x"][0], b["x"][1].to(dev)
    img = b["img"].to(dev)
    g = torch.Generator().manual_seed(5)
    n = locs.shape[0]
    up3 = [torch.randn(n, 5, generator=g).to(dev) for _ in range(2)]
    up2 = [torch.randn(n, 5, generator=g).to(dev) for _ in range(2)]

    def run(model, batch, ups, sync):
        model.zero_grad(set_to_none=True)
        if sync:
            syncbn.enable()
            assert syncbn.active()
            out = model(batch)
        else:
            with syncbn.local_statistics():
                out = model(batch)
                (out["seg_logit"] * ups[0]).sum().add((out["seg_logit2"] * ups[1]).sum()).backward()
            return out
        (out["seg_logit"] * ups[0]).sum().add((out["seg_logit2"] * ups[1]).sum()).backward()
        return out

    for tag, build, batch, ups in (("3d", build_model_3d, {"x": [locs, feats]}, up3),
                                   ("2d", build_model_2d, {"img": img, "img_indices": b["img_indices"]}, up2)):
        ms, ml = build(cfg)[0].to(dev).train(), build(cfg)[0].to(dev).train()
        sd = {k: det_tensor(k, tuple(v.shape)) for k, v in ms.state_dict().items()}
        ms.load_state_dict(sd)
        ml.load_state_dict(sd)
        for m in (ms, ml):
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
        o_l = run(ml, batch, ups, sync=False)
        o_s = run(ms, batch, ups, sync=True)
        syncbn.disable()
        for k in ("seg_logit", "seg_logit2"):
            err = float((o_s[k] - o_l[k]).abs().max()) / max(1.0, float(o_l[k].abs().max()))
            if not err <= 2e-5:
                failures.append(f"{tag} {k}: one-rank synchronised BatchNorm differs from rank-local BatchNorm by {err:.2e}")
        num = den = 0.0
        for (nm, ps), (_, pl) in zip(ms.named_parameters(), ml.named_parameters()):
            if pl.grad is None:
                continue
            num += float((ps.grad.double() - pl.grad.double()).square().sum())
            den += float(pl.grad.double().square().sum())
        rel = (num / den) ** 0.5
        print(f"[rccl] {tag}: one-rank SyncBN over nccl vs local BatchNorm: gradient L2 difference {rel:.2e}", file=sys.stderr, flush=True)
        if not rel <= 5e-3:
            failures.append(f"{tag}: gradient L2 difference {rel:.2e}")
        for k, v in ms.state_dict().items():
            if "running_" in k:
                ref = ml.state_dict()[k]
                if not float((v - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())):
                    failures.append(f"{tag} buffer {k} differs")

        # the flat gradient buffer through RCCL: a one-rank sum leaves it unchanged; asynchronously on a side stream as bench.py does
        opt = FlatAdam(ml.parameters(), lr=1e-3)
        opt.zero_grad()
        o = ml(batch)
        (o["seg_logit"] * ups[0]).sum().backward()
        before = opt.grad.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            work = opt.all_reduce(async_op=True)
            assert work is not None, "FlatAdam.all_reduce did not issue a collective under MOPA_FORCE_COLLECTIVES=1"
            work.wait()
            opt.step(1.0)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        if opt.n_collectives != 1 or opt.collective_backend != "nccl":
            failures.append(f"{tag}: collectives {opt.n_collectives} via {opt.collective_backend}")
        if not torch.equal(opt.grad, before):
            failures.append(f"{tag}: a one-rank all-reduce changed the flat gradient buffer")
        if not torch.isfinite(opt.flat).all():
            failures.append(f"{tag}: parameters not finite after the update")

        # the same gradient through 3 buckets issued from inside the backward pass on the communication stream (FlatAdam.enable_buckets):
        # every bucket goes out before backward() returns, and a one-rank sum leaves the buffer equal to the un-bucketed gradient
        opt.enable_buckets(3)
        opt.zero_grad()
        o = ml(batch)
        (o["seg_logit"] * ups[0]).sum().backward()      # un-bucketed reference at the updated weights
        ref = opt.grad.clone()
        opt.zero_grad()
        n0 = opt.n_collectives
        o = ml(batch)
        loss_b = (o["seg_logit"] * ups[0]).sum()
        if not opt.arm_buckets():
            failures.append(f"{tag}: arm_buckets refused under a process group")
        loss_b.backward()
        early, left = sum(opt._bucket_issued), list(opt._bucket_left)
        opt.all_reduce()
        torch.cuda.synchronize()
        if early != 3 or opt.n_collectives - n0 != 3:
            failures.append(f"{tag}: {early} of 3 buckets issued inside the backward pass (gradients still missing per bucket: {left}), "
                            f"{opt.n_collectives - n0} collectives")
        if not torch.equal(opt.grad, ref):
            failures.append(f"{tag}: the bucketed all-reduce changed the flat gradient buffer")
    dist.barrier()
    dist.destroy_process_group()
    if failures:
        print("\n".join(failures), file=sys.stderr)
        sys.exit(1)
    print("[rccl] ok", file=sys.stderr)


if __name__ == "__main__":
    main()
