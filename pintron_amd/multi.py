"""est-fact over several GPUs of one node: one rank per GPU, ESTs split in contiguous ranges.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m pintron_amd.multi <directory with genomic.txt and ests.txt>

Every rank runs the est-fact program (pintron_amd/host, through libestfact.so) on its range with the
genomic sequence and its index replicated; rank 0 gathers the text of the six output files over RCCL
and leaves in <directory> exactly the files a single est-fact process writes (SURVEY.md section 8e;
pintron_amd/estfact.py: run_sharded).  torch is imported before the library is loaded, see
INTEGRATION.md section 5.  PINTRON_DIST_BACKEND=gloo (with PINTRON_ESTFACT_LIB pointing at the check
build) runs the same driver on CPU for the tests.
"""
import os
import shutil
import sys
import tempfile


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit(__doc__)
    directory = os.path.abspath(argv[0])
    import torch
    import torch.distributed as dist
    from . import estfact
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PINTRON_DIST_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(local)
        os.environ["PINTRON_GPU_DEVICE"] = str(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        device = "cuda"
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
        device = "cpu"
    work = tempfile.mkdtemp(prefix="pintron_rank%d_" % rank)
    try:
        stats = estfact.run_sharded(directory, work, dist, rank, world, device, files=(0, 1, 2, 3, 4, 5))
        if os.environ.get("PINTRON_VERBOSE"):
            print("rank %d: %d ESTs, %d aligned, %d DP jobs" % (rank, stats["ests"], stats["aligned"], stats["dp_jobs"]),
                  file=sys.stderr)
        dist.barrier()
    finally:
        shutil.rmtree(work, ignore_errors=True)
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
