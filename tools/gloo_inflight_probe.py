"""Round-1 rehearsal hang (gpurun_out/rehearse4g.log: 4 gloo ranks sharing one GPU, all stuck in GradBucketReducer.wait)
reproduced in isolation: N ranks on cuda:0, gloo over DEVICE tensors, K buckets, every async work left outstanding
(max_inflight = 0) vs capped at 2.  Prints one line per configuration; a configuration that does not finish within
its time limit prints HANG with the stacks (faulthandler) and is killed.  Usage: python tools/gloo_inflight_probe.py"""
import faulthandler, os, socket, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, inflight, nbuckets, device):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from multimodal_unlearning_amd.engine import GradBucketReducer

    faulthandler.dump_traceback_later(40, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device(device)
    n = nbuckets * (1 << 20)
    flat = torch.full((n,), float(rank + 1), device=dev, dtype=torch.bfloat16)
    red = GradBucketReducer(flat, bucket_elems=1 << 20, max_inflight=inflight)
    t0 = time.time()
    for step in range(3):
        flat.fill_(float(rank + 1))
        # hooks arrive top-down in a few ranges, like the model's sweep
        q = n // 4
        for lo, hi in ((3 * q, n), (2 * q, 3 * q), (q, 2 * q), (0, q)):
            red.mark_final(lo, hi)
        red.reduce_rest()
        red.wait()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        exp = world * (world + 1) / 2
        assert float(flat.float().mean()) == exp, (float(flat.float().mean()), exp)
    if rank == 0:
        print(f"OK world={world} device={device} buckets={nbuckets} max_inflight={inflight}: 3 steps in {time.time() - t0:.1f}s", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(*[int(x) for x in sys.argv[1:6]], sys.argv[6])
        sys.exit(0)
    for device in ("cpu", "cuda:0"):
        for world, inflight in ((4, 2), (4, 0), (3, 0)):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            ps = [subprocess.Popen([sys.executable, __file__, str(r), str(world), str(port), str(inflight), "16", device])
                  for r in range(world)]
            t0, hung = time.time(), False
            for p in ps:
                try:
                    p.wait(timeout=max(1, 90 - (time.time() - t0)))
                except subprocess.TimeoutExpired:
                    hung = True
            if hung or any(p.returncode != 0 for p in ps):
                for p in ps:
                    if p.poll() is None:
                        p.kill()
                print(f"HANG/FAIL world={world} device={device} max_inflight={inflight} rc={[p.returncode for p in ps]}", flush=True)
