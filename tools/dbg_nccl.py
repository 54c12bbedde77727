"""GPU box: does pgpu_init still work after torch + RCCL are up, for either library load order?"""
import os, sys, ctypes as C, subprocess
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29656")
order = sys.argv[1]
if order == "lib_first":
    import pintron_amd.capi as capi
    L = capi.lib()
import torch, torch.distributed as dist
import pintron_amd.capi as capi
L = capi.lib()
def try_init(tag):
    ctx = C.c_void_p()
    rc = L.pgpu_init(0, C.byref(ctx))
    print(tag, "pgpu_init rc", rc, flush=True)
try_init("before torch.cuda")
torch.cuda.set_device(0)
x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
try_init("after torch.cuda init")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try_init("after nccl init_process_group")
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
try_init("after first collective")
print(subprocess.run("cat /proc/%d/maps | grep -o '/[^ ]*libamdhip64[^ ]*\|/[^ ]*libhsa-runtime[^ ]*' | sort | uniq -c" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
dist.destroy_process_group()
