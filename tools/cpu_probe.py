import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import bench
for thr in (16, 32, 64):
    os.environ["HAMT_CPU_THREADS"] = str(thr)
    t=time.time()
    torch.set_num_threads(thr)
    from oracle.hamt_oracle import HamtOracle, OracleConfig, make_state_dict, pretrain_param_shapes
    from vln_hamt_amd.synth import make_batch
    cfg = OracleConfig()
    if thr == 16:
        sd = make_state_dict(pretrain_param_shapes(cfg), seed=1)
        params = {k: v.requires_grad_(True) for k, v in sd.items() if k != "mlm_head.predictions.decoder.weight"}
        print("sd built", time.time()-t, flush=True)
    for i in range(3):
        b = make_batch("sap", 16, cfg, seed=i)
        t0=time.time()
        loss = HamtOracle(params, cfg, training=True).forward(b, "sap", True).mean()
        loss.backward()
        for p in params.values(): p.grad=None
        print(thr, "step", i, time.time()-t0, flush=True)
