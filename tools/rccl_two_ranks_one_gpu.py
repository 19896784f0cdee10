#!/usr/bin/env python3
"""Developer probe: can two ranks of a native RCCL communicator share the box's one GPU?  (NCCL refuses duplicate GPUs;
whatever happens must be a clean status, never a hang: run under `timeout`.)"""
import multiprocessing as mp, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(rank, path, q):
    from camera_intrinsic_calibration_rs_amd import engine
    from camera_intrinsic_calibration_rs_amd.engine import Context
    if rank == 0:
        uid = engine.rccl_unique_id()
        open(path + ".tmp", "wb").write(uid); os.rename(path + ".tmp", path)
    else:
        while not os.path.exists(path): time.sleep(0.01)
        uid = open(path, "rb").read()
    ctx = Context(0)
    try:
        comm = ctx.rccl_comm_create(2, rank, uid)
        q.put((rank, "created", comm is not None))
    except Exception as e:
        q.put((rank, "error", repr(e)))

if __name__ == "__main__":
    mp.set_start_method("spawn")
    q = mp.Queue(); path = os.path.join(tempfile.mkdtemp(), "id")
    ps = [mp.Process(target=worker, args=(r, path, q)) for r in range(2)]
    for p in ps: p.start()
    for _ in range(2):
        try: print(q.get(timeout=60))
        except Exception as e: print("no answer", repr(e))
    for p in ps:
        p.join(timeout=5)
        if p.is_alive(): p.kill()
