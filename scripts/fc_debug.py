"""Tuning aid: MFMA MLP vs the VALU kernel with parts of the FC weights zeroed (which part of the data flow differs)."""
import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")): sys.path.insert(0, p)
    import numpy as np, torch
    import convert_weights
    from art_planner_amd.context import Context
    from synthetic import raw_map
    dev = torch.device("cuda", 0)
    g = raw_map(400, 0.04, seed=1234)
    elv = np.ascontiguousarray(g["elevation"][::-1, ::-1]).astype(np.float32)
    rng = np.random.default_rng(3)
    n = 4096
    s = rng.uniform(-7.5, 7.5, (n, 2)); d = rng.uniform(-0.6, 0.6, (n, 2))
    e = np.stack([s[:, 0] + d[:, 0], s[:, 1] + d[:, 1], rng.uniform(-np.pi, np.pi, n), s[:, 0], s[:, 1], rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
    TOTAL = 160+16+48*64+48+24*48+24+24*48+24+36*48+36+24+1+24+1+36+1
    base = bytearray(convert_weights.to_blob(convert_weights.random_params(0)))
    res = []
    for case in ("full", "no_t_inputs", "only_t_inputs", "no_bias0", "heads_k_lt_32", "heads_k_ge_32"):
        b = bytearray(base)
        w = np.frombuffer(bytes(b[-4 * TOTAL:]), dtype=np.float32).copy()
        W0 = w[176:176 + 48 * 64].reshape(48, 64)
        if case == "no_t_inputs": W0[:, 48:] = 0
        if case == "only_t_inputs": W0[:, :48] = 0
        if case == "no_bias0": w[176 + 48 * 64:176 + 48 * 64 + 48] = 0; W0[:, 48:] = 0
        o = 176 + 48 * 64 + 48
        for nn in (24, 24, 36):
            H = w[o:o + nn * 48].reshape(nn, 48)
            Hb = w[o + nn * 48:o + nn * 48 + nn]
            if case == "heads_k_lt_32": H[:, 32:] = 0
            if case == "heads_k_ge_32": H[:, :32] = 0
            if nn == 36:
                if case == "h3_rows_0_15": H[16:] = 0; Hb[16:] = 0
                if case == "h3_rows_16_31": H[:16] = 0; H[32:] = 0; Hb[:16] = 0; Hb[32:] = 0
                if case == "h3_rows_32_35": H[:32] = 0; Hb[:32] = 0
                if case == "h3_no_bias": Hb[:] = 0
                if case == "h3_rows_0_3": H[4:] = 0; Hb[4:] = 0
            o += nn * 48 + nn
        b[-4 * TOTAL:] = w.tobytes()
        ctx = Context(0, "yaml"); ctx.use_torch_stream()
        ctx.cost_set_fc_path(os.environ.get("ARTP_FC_MFMA", "1") != "0")   # the script's own switch -> the setter
        ctx.cost_load_weights(bytes(b))
        ctx.cost_update_map(elv, g.res, g.len_x, g.len_y)
        res.append(ctx.cost_query(e))
        ctx.close()
    np.save(sys.argv[2], np.stack(res))
else:
    import numpy as np
    r = {}
    for tag, env in (("mfma", {}), ("valu", {"ARTP_FC_MFMA": "0"})):
        f = f"/tmp/fc_dbg_{tag}.npy"
        subprocess.call([sys.executable, os.path.abspath(__file__), "child", f], env=dict(os.environ, **env))
        r[tag] = np.load(f)
    for i, case in enumerate(("full", "no_t_inputs", "only_t_inputs", "no_bias0", "heads_k_lt_32", "heads_k_ge_32")):
        d = np.abs(r["mfma"][i] - r["valu"][i])
        print(f"{case:16s} max {d.max(0)} mean {d.mean(0)}  |valu| mean {np.abs(r['valu'][i]).mean(0)}")
        print("   mfma", np.round(r["mfma"][i][:6], 4).tolist())
        print("   valu", np.round(r["valu"][i][:6], 4).tolist())
