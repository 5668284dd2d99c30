#!/usr/bin/env python3
"""N-rank driver of the record-sharded batches (BASELINE configs[3] / configs[4] and the many-target-grids broadcast).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/run_sharded.py \
        --case cfg4|cfg5|targets [--fields 256] [--size full|small] [--verify]

One process per GPU over RCCL ("nccl").  With a single process (`python tools/run_sharded.py ...`) no process group is
created.  --verify re-computes every record of rank 0 through the plain single-field entry points (c_ezsint_dev,
ezhip_pack16_compress_dev) and compares checksums / byte counts: the sharded drivers return exactly those records.
Rank 0 prints one JSON line."""
import argparse, json, os, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", choices=["cfg4", "cfg5", "targets"], default="cfg4")
    ap.add_argument("--fields", type=int, default=64)
    ap.add_argument("--size", choices=["full", "small"], default="small")
    ap.add_argument("--chunk", type=int, default=32)
    ap.add_argument("--verify", action="store_true")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    torch.cuda.set_device(local)                      # before anything touches the GPU
    if world > 1 or os.environ.get("RUN_SHARDED_FORCE_DIST"):
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from librmn_amd import batch_driver as bd
    import ezcases as ec
    be = bd.HipBackend()
    src = (4400, 2200, "G", 0, 0, 0, 0) if args.size == "full" else (360, 181, "G", 0, 0, 0, 0)
    dst = (7200, 3601, "L", 5, 5, 0, 0) if args.size == "full" else (520, 261, "L", 69, 69, 0, 0)
    # the source records of this rank, resident in HBM BEFORE anything is timed (a record is a device-side variation of one synthetic base field:
    # building every field with numpy inside the timed region measured the host, 4.8 fields/s at full size)
    from librmn_amd import sharding as sh
    base = be.to_device(ec.synth_field(src[0], src[1], seed=1000))
    mine = sh.fields_of_rank(args.fields, rank, world) if args.case != "targets" else []
    fields = {f: (base * (1.0 + 1e-4 * (f % 97)) + 0.01 * f) for f in mine}
    be.sync()
    make = fields.__getitem__
    out = {"case": args.case, "world": world, "fields": args.fields, "size": args.size}
    if args.case in ("cfg4", "cfg5"):
        h = be.define_set(src, dst)
        res = (bd.run_interp_batch if args.case == "cfg4" else bd.run_pack_batch)(be, h, args.fields, make, rank, world, args.chunk, comm_device="cuda")
        out.update({k: res[k] for k in res if k != "records_of_rank"})
        out["fields_per_s"] = args.fields / res["seconds"]
        if args.verify and rank == 0:
            from librmn_amd import ezscint as ez, packers as pk
            ok = True
            for f in res["records_of_rank"][:8]:
                d_in = make(f)
                z = torch.empty(h["nout"], dtype=torch.float32, device="cuda")
                assert ez.ezsint_dev(z, d_in) in (0, 2)
                if args.case == "cfg4":
                    ok = ok and be.checksum(z) == res["checksums"][f]
                else:
                    rec = torch.zeros(bd.record_stride_words(h["nout"]), dtype=torch.int32, device="cuda")
                    zl = pk.pack16_compress_dev(rec, z, dst[0], dst[1], 16)
                    torch.cuda.synchronize()
                    nb = 16 + ((zl - 1) // 4) * 4 if zl > 0 else 16 + 2 * h["nout"]
                    ok = ok and zl == res["zlng"][f] and be.checksum(rec, nb) == res["checksums"][f]
            out["verified_against_single_field_calls"] = bool(ok)
    else:
        targets = [(90 + 30 * t, 46 + 15 * t, "L", 400 - 20 * t, 400 - 20 * t, 0, 0) for t in range(8)] if args.size == "small" else \
                  [(7200, 3601, "L", 5, 5, 0, 0), (3600, 1801, "L", 10, 10, 0, 0), (1440, 721, "L", 25, 25, 0, 0), (720, 361, "L", 50, 50, 0, 0)]
        mk = lambda: be.to_device(ec.synth_field(src[0], src[1], seed=7)) if rank == 0 else torch.zeros(src[0] * src[1], dtype=torch.float32, device="cuda")   # noqa: E731
        res = bd.run_many_targets(be, src, targets, mk, rank, world, comm_device="cuda")
        out.update({"checksums": res["checksums"], "targets": len(targets)})
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or os.environ.get("RUN_SHARDED_FORCE_DIST"):
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
