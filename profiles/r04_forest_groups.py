"""The config-5 forest of bench.py as the forest kernel sees it: nodes per tree, the LDS groups rsseg_forest_load forms
(<= 4 consecutive trees whose nodes fit the node area beside 1024 feature rows), and what a workgroup moves per group.
python profiles/r04_forest_groups.py  (on the GPU box; prints JSON)"""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT]
import numpy as np, torch, bench
from rsseg import pipeline as P
dev = torch.device("cuda:0")
fm = bench.fit_c5_forest(torch, None, dev, P, 0, 1, 16384)
nn = [int(e.tree_.node_count) for e in fm["model"].estimators_]
F, TH = 19, 1024
cap = min((160 * 1024 - 256 - (F | 1) * TH * 4 - 16) // 8, 2 * 6 * TH) & ~1
groups, t, off = [], 0, 0
offs = np.concatenate([[0], np.cumsum(nn)])
while t < len(nn):
    base = int(offs[t]) & ~1
    cnt = 0
    while t < len(nn) and cnt < 4 and offs[t] + nn[t] - base <= cap:
        cnt += 1
        t += 1
    assert cnt > 0
    groups.append(cnt)
print(json.dumps({"trees": len(nn), "nodes_per_tree": {"min": min(nn), "median": int(np.median(nn)), "max": max(nn), "total": int(sum(nn))},
                  "lds_node_capacity": int(cap), "groups": len(groups), "trees_per_group": {str(k): groups.count(k) for k in sorted(set(groups))},
                  "max_depth": [int(e.tree_.max_depth) for e in fm["model"].estimators_][:8]}))
