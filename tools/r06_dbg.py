"""round 6 debugging aid: one column-sharded window-mode case on ranks sharing the GPU, with the error message of every rank"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _mp_launch import run_world
world = int(sys.argv[1]); a = json.loads(sys.argv[2])
res = run_world("gpu_crs", dict(a, params=dict({"amd_cu_share": world}, **(a.get("params") or {})), want_errmsg=True), world=world)
for r, d in enumerate(res):
    print("rank", r, "ret", int(d["ret"][0]), "nevals", int(d["nevals"][0]), "rounds", int(d["rounds"][0]), "collectives", int(d["collectives"][0]), "msg:", str(d.get("errmsg")))
