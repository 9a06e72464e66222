#!/usr/bin/env python3
"""End-to-end check of the multi-rank RHS path (RhsEngine + HaloExchanger) against a single engine, bit for bit.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node R --master-addr 127.0.0.1 --master-port P \
      tools/check_sharded.py [--backend nccl|gloo] [--formulation cns|euler|hex]

With --backend gloo all ranks may share one GPU (traces staged through the host); nccl needs one GPU per rank."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from common import (becker_constants, product_cavity_problem, product_cns_problem, product_euler_problem,  # noqa: E402
                    product_hex_problem, product_shocktube_problem)
from esdg_cns_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="gloo")
ap.add_argument("--formulation", default="cns")
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr % torch.cuda.device_count())
if args.backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", lr % torch.cuda.device_count()))
else:
    dist.init_process_group(args.backend)

if args.formulation == "hex":
    N, dims = 3, (4, 4, 4 * world)          # dyadic spacings for world = 2, 4: bitwise comparable geometry
    layer = dims[0] * dims[1]
    offs = [layer * 4 * r for r in range(world + 1)]
    build = lambda er: product_hex_problem(N, *dims, elem_range=er)
    form, kw = engine.EULER_HEX_COLLOCATED, dict(lf_scale=0.25)
elif args.formulation in ("cavity", "shocktube"):
    N, Kx, Ky = 3, 8, 4 * world
    offs = [Kx * 4 * r for r in range(world + 1)]
    if args.formulation == "cavity":                      # walls + lid, BCTYPE 2, penalty on
        build = lambda er: product_cavity_problem(N, Kx, Ky, elem_range=er)
        form, kw = engine.CNS_MODAL, dict(BCTYPE=2)
    else:                                                 # Dirichlet inflow / copy outflow, periodic in y (the sharded direction)
        st = becker_constants()
        build = lambda er: product_shocktube_problem(N, Kx, Ky, elem_range=er)
        form, kw = engine.CNS_MODAL, dict(BCTYPE=4, viscous_dissp=False, mu=st["mu"], lam=st["lam"], Pr=st["Pr"],
                                          inflow=(st["rhoL"], st["uL"], st["vL"], st["pL"]))
else:
    N, Kx, Ky = 4, 12, 3 * world
    offs = [Kx * 3 * r for r in range(world + 1)]
    build = (lambda er: product_cns_problem(N, Kx, Ky, elem_range=er)) if args.formulation == "cns" else \
            (lambda er: product_euler_problem(N, Kx, Ky, elem_range=er))
    form, kw = (engine.CNS_MODAL if args.formulation == "cns" else engine.EULER_COLLOCATED), {}

rd, md, ops, Q = build((offs[rank], offs[rank + 1]))
eng = engine.RhsEngine(rd, md, ops, form, rank=rank, nranks=world, rank_offsets=offs, **kw)
Qd = eng.upload(Q)
out = eng.rhs(Qd)
out2 = eng.rhs(Qd)                       # second evaluation: exchanges are re-entrant
assert torch.equal(out, out2)
# fused RK stage through the sharded path
res = torch.full_like(Qd, 0.01)
Q1 = Qd.clone()
eng.rhs_lsrk_fused(Q1, res, -0.41789, 0.37921, 1e-3)
torch.cuda.synchronize()
parts = [None] * world
dist.all_gather_object(parts, (out.cpu().numpy(), Q1.cpu().numpy()))
# error functionals reduce their partial sums over the ranks
from esdg_cns_amd import setup_dg as sd  # noqa: E402
func = None
if args.formulation in ("cns", "euler"):
    Vq2, wq2 = sd.error_quadrature(N)
    eng.setup_errors(rd, md, Vq2, wq2)
    func = eng.l2_error(Qd, 0.1)[0]
elif args.formulation == "cavity":
    eng.setup_errors(rd, md, boundary=True)
    func = eng.boundary_velocity_error(Qd, 0.25)[1]
ok = True
if rank == 0:
    rdf, mdf, opsf, Qf = build(None)
    full = engine.RhsEngine(rdf, mdf, opsf, form, **kw)
    Qfd = full.upload(Qf)
    Qfd0 = Qfd.clone()
    ref = full.rhs(Qfd).cpu().numpy()
    resf = torch.full_like(Qfd, 0.01)
    full.rhs_lsrk_fused(Qfd, resf, -0.41789, 0.37921, 1e-3)
    got = np.concatenate([p[0] for p in parts], axis=1)
    gotQ = np.concatenate([p[1] for p in parts], axis=1)
    bitwise = np.array_equal(got, ref) and np.array_equal(gotQ, Qfd.cpu().numpy())
    rel = np.abs(got - ref).max() / np.abs(ref).max()
    # the shard's geometry comes from its own host set-up; with mesh spacings that are not dyadic the BLAS products
    # there may differ from the full mesh's by an ulp, hence the round-off fallback
    ok = bitwise or rel < 1e-12
    if func is not None:
        if args.formulation == "cavity":
            full.setup_errors(rdf, mdf, boundary=True)
            fref = full.boundary_velocity_error(Qfd0, 0.25)[1]
        else:
            full.setup_errors(rdf, mdf, Vq2, wq2)
            fref = full.l2_error(Qfd0, 0.1)[0]
        ok = ok and abs(func - fref) <= 1e-12 * abs(fref)
        print(f"  error functional sharded {func:.15e} vs single {fref:.15e}")
    lo, hi = eng.interior
    print(f"check_sharded {args.formulation} world={world} backend={args.backend}: "
          f"{'BITWISE EQUAL' if bitwise else ('EQUAL TO ROUND-OFF' if ok else 'MISMATCH')} "
          f"(max rel diff {rel:.3e}; interior [{lo},{hi}) of {eng.K}, overlap schedule "
          f"{'on' if eng.overlap and hi > lo else 'off'})")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
