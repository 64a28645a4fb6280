"""RCCL first contact on ONE GPU (no multi-GPU node has been available to any round): a world of one rank on the "nccl" backend,
driven through the very helpers the N > 1 paths use - cbas_amd.dist.init_from_env / barrier / max_over_ranks, an all_gather of
row counts as gather_rows posts it, and the point-to-point form of encode_files (_p2p + _wait_done: a group of one send and a
group of one receive on N > 1; here, to / from rank 0 itself, one group holding both - RCCL matches a self send only inside its
group - posted from a second thread like rank 0's receiver thread) on buffers that kernels of the library's own streams rewrite afterwards.  What it can show: the RCCL library
loads and builds a communicator under HSA_ENABLE_IPC_MODE_LEGACY=0, collectives and grouped point-to-point calls complete,
Work.is_completed() behaves as _wait_done assumes, and RCCL kernels co-exist with the encoder's streams.  What it cannot: xGMI,
two processes, IPC handles.     usage: python scripts/rccl_first_contact.py [out.json]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ["WORLD_SIZE"] = "1"
os.environ["RANK"] = "0"
os.environ["LOCAL_RANK"] = "0"
# the HSA runtime reads this when it starts - i.e. at the first device call below, so it must be set BEFORE torch touches the
# GPU (it used to be set after torch.cuda.set_device(0), where it had no effect on a stand-alone run: ADVICE r4)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from cbas_amd import dist as cdist  # noqa: E402


def main() -> None:
    out = {"steps": []}

    def step(name, fn):
        t0 = time.perf_counter()
        val = fn()
        out["steps"].append({"step": name, "seconds": round(time.perf_counter() - t0, 4), "result": val})
        print(f"{name}: {val}  ({out['steps'][-1]['seconds']} s)", flush=True)

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # init_from_env keeps a world of one un-initialised (the product never needs a group then); here the group is the point
    step("init_process_group(nccl, world 1)", lambda: (dist.init_process_group(backend="nccl", rank=0, world_size=1), dist.get_backend())[1])
    step("barrier (creates the communicator)", lambda: (cdist.barrier(), "ok")[1])

    def allreduce():
        t = torch.tensor([3.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    step("all_reduce MAX (max_over_ranks' collective)", allreduce)

    def allgather():
        meta = torch.tensor([2, 768, 0, 4096, 17], dtype=torch.int64, device=dev)
        got = [torch.zeros_like(meta)]
        dist.all_gather(got, meta)
        return got[0].cpu().tolist()
    step("all_gather of a row-count block (gather_rows' control step)", allgather)

    # the encode_files transfers: rows leave from device buffers that later kernels rewrite; the receive is posted from another
    # thread.  World 1: both ends are rank 0 (RCCL matches a send and a receive to self inside one communicator).
    from cbas_amd import config as C, weights as W, synth
    from cbas_amd.encoder import DinoEncoder
    cfg = C.VIT_TINY
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), dev, max_batch=16, max_frame=(64, 64))
    frames = torch.from_numpy(synth.cage_frames(9, 16, 64, 64)).to(dev)
    rows16, _ = enc.encode_u8(frames)
    torch.cuda.synchronize()
    want = rows16.clone()
    big = torch.randn((18000, 768), device=dev).to(torch.float16)          # one cfg3 clip's rows: 27.6 MB
    got_small = torch.empty_like(rows16)
    got_big = torch.empty_like(big)
    box = {}

    def transfers():
        # RCCL matches a send to self only with a receive posted in the SAME group, so each clip is one group of two
        try:
            torch.cuda.set_device(dev)
            for src, dst in ((rows16, got_small), (big, got_big)):
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]):
                    cdist._wait_done(w)
            box["p2p"] = "ok"
        except BaseException as e:  # noqa: BLE001
            box["p2p"] = f"{type(e).__name__}: {e}"

    def p2p():
        th = threading.Thread(target=transfers, name="cbas-gather", daemon=True)
        th.start()
        for _ in range(8):                      # the encode loop goes on (the library's own streams) while the transfers run
            enc.encode_u8(frames)
        th.join(60)
        if th.is_alive():
            return {"p2p": "did not return within 60 s", "bytes_identical": False}
        torch.cuda.synchronize()
        same = box.get("p2p") == "ok" and bool(torch.equal(got_small, want)) and bool(torch.equal(got_big, big))
        return {"p2p": box.get("p2p"), "bytes_identical": same}
    step("grouped isend + irecv to self from a second thread (the receiver thread's form), encoder running beside it", p2p)

    def rate():
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, big, 0), dist.P2POp(dist.irecv, got_big, 0)]):
                cdist._wait_done(w)
        dt = (time.perf_counter() - t0) / n
        return {"ms_per_27.6MB_clip": round(dt * 1e3, 3), "GB_s": round(big.numel() * 2 / dt / 1e9, 1)}
    step("self transfer rate (HBM to HBM through RCCL's kernels; not xGMI)", rate)

    enc.close()
    step("destroy_process_group", lambda: (dist.destroy_process_group(), "ok")[1])
    out["ok"] = all("did not" not in str(s["result"]) for s in out["steps"]) and out["steps"][4]["result"].get("bytes_identical") is True
    out["versions"] = {"torch": torch.__version__, "nccl": ".".join(map(str, torch.cuda.nccl.version())) if hasattr(torch.cuda, "nccl") else None}
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out))
    if not out["ok"]:
        sys.exit(1)


if __name__ == "__main__":
    main()
