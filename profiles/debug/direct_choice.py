import sys, os
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")]
import importlib
capi = importlib.import_module("deep-fusion_amd.capi")
import bench
for wl in ("vgg5", "vgg3"):
    case, desc = bench.workloads()[wl]
    print(wl, case)
    for env in ({}, {"DFX_DIRECT_NW": "8", "DFX_DIRECT_NPB": "2"}):
        for k, v in env.items(): capi.lib().dfx_debug_set_tuning(k.encode(), v.encode())
        op = capi.Conv((case.bs, case.ih, case.iw, case.ic), (case.oc, case.ic, 3, 3), dst_dt=case.dst_dt, oc1x1=0)
        i = op.info()
        print(env, i.kernel_name.decode(), "grid", i.grid, "rows/unit", i.rows_per_unit, "lds", i.lds_bytes)
        op.close()
        for k in env: capi.lib().dfx_debug_set_tuning(k.encode(), None)
