"""Tuning aid: the cost query (R9) with the MFMA MLP and with the fp32 VALU kernels (artp_cost_set_fc_path(ctx, 0) -- $ARTP_FC_MFMA=0 is this script's own switch for it, own process):
time per batch (HIP events) at 50 000 and 2^20 edges, and the costs themselves for a comparison."""
import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")): sys.path.insert(0, p)
    import numpy as np, torch
    import convert_weights
    from art_planner_amd.context import Context
    from synthetic import raw_map
    dev = torch.device("cuda", 0)
    ctx = Context(0, "yaml"); ctx.use_torch_stream()
    ctx.cost_set_fc_path(os.environ.get("ARTP_FC_MFMA", "1") != "0")   # the script's own switch -> the setter
    ctx.cost_load_weights(convert_weights.to_blob(convert_weights.random_params(0)))
    g = raw_map(400, 0.04, seed=1234)
    elv = np.ascontiguousarray(g["elevation"][::-1, ::-1]).astype(np.float32)
    ctx.cost_update_map(elv, g.res, g.len_x, g.len_y)
    rng = np.random.default_rng(3)
    out = {}
    for n in (50000, 1 << 20):
        s = rng.uniform(-8.5, 8.5, (n, 2)); d = rng.uniform(-0.6, 0.6, (n, 2))
        e = np.stack([s[:, 0] + d[:, 0], s[:, 1] + d[:, 1], rng.uniform(-np.pi, np.pi, n), s[:, 0], s[:, 1], rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
        et = torch.from_numpy(e).to(dev); ct = torch.empty((n, 3), dtype=torch.float32, device=dev)
        for _ in range(3): ctx.cost_query_dev(et, ct)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ctx.cost_query_dev(et, ct)
        e1.record(); torch.cuda.synchronize()
        out[n] = (e0.elapsed_time(e1) / 20, ct.cpu().numpy())
        print(f"  {n:8d} edges: {out[n][0] * 1e3:8.1f} us  = {n / out[n][0] / 1e6:8.2f} G queries/s", flush=True)
    np.save(sys.argv[2], np.concatenate([out[50000][1], out[1 << 20][1]]))
else:
    import numpy as np
    res = {}
    for tag, env in (("mfma", {}), ("valu", {"ARTP_FC_MFMA": "0"})):
        print(tag, flush=True)
        f = f"/tmp/fc_ab_{tag}.npy"
        subprocess.call([sys.executable, os.path.abspath(__file__), "child", f], env=dict(os.environ, **env))
        res[tag] = np.load(f)
    d = np.abs(res["mfma"] - res["valu"])
    print("max |mfma - valu| per output", d.max(0), "mean", d.mean(0), "max relative", (d / (np.abs(res["valu"]) + 1e-3)).max())
