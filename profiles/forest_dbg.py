import sys, os, copy, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "rs-image-segmentation_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "profiles")]
from oracle import ref_np as oracle
from rsseg.runtime import Context
from forest_repro_case import case
ctx = Context(0)
dev = lambda a: ctx.to_device(np.ascontiguousarray(a).reshape(-1))
for seed in (21, 62, 90):
    model, X, tag = case(seed)
    F = X.shape[1]
    planes = [dev(X[:, i]) for i in range(F)]
    f = oracle.flatten_forest(model)
    ctx.forest_load(f)
    got = ctx.forest_predict(planes).cpu().numpy()
    want = oracle.rf_predict_planes(f, [np.ascontiguousarray(X[:, i]) for i in range(F)])
    print(tag, "gpu-vs-oracle differing", int((got != want).sum()), flush=True)
    for t, est in enumerate(model.estimators_):
        m1 = copy.copy(model)
        m1.estimators_ = [est]
        m1.n_estimators = 1
        f1 = oracle.flatten_forest(m1)
        ctx.forest_load(f1)
        g1 = ctx.forest_predict(planes).cpu().numpy()
        w1 = oracle.rf_predict_planes(f1, [np.ascontiguousarray(X[:, i]) for i in range(F)])
        bad = np.where(g1 != w1)[0]
        if bad.size:
            tr = est.tree_
            i = bad[0]
            print("  tree", t, "nodes", tr.node_count, "differing", bad.size, "row", i, "gpu", g1[i], "oracle", w1[i])
            print("   feature", tr.feature[:15].tolist())
            print("   threshold", tr.threshold[:15].tolist())
            print("   left", tr.children_left[:15].tolist(), "right", tr.children_right[:15].tolist())
            print("   missing_go_to_left", getattr(tr, "missing_go_to_left", None)[:15].tolist() if hasattr(tr, "missing_go_to_left") else None)
            print("   x", X[i][[int(a) for a in tr.feature[:15] if a >= 0]].tolist(), "leaf(sklearn)", int(tr.apply(X[i:i+1])[0]))
            print("   value rows", tr.value[:15, 0, :].round(3).tolist())
            break
