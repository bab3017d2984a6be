import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "aule-attention_amd"))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "libhsa-runtime" in l})
if order == "aule_first":
    from aule import _capi
    lib = _capi.load()
    print("after libaule:", maps())
    import torch
    print("after torch:", maps(), "torch sees", torch.cuda.is_available())
    print("aule_init ->", lib.aule_init(), _capi.last_error(lib) if False else "")
else:
    import torch
    print("after torch:", maps(), "torch sees", torch.cuda.is_available())
    from aule import _capi
    lib = _capi.load()
    print("after libaule:", maps())
    print("aule_init ->", lib.aule_init())
