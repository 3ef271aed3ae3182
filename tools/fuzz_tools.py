#!/usr/bin/env python3
"""A long fuzz of the other plan-level kernels of the path against the oracle, on random shapes (1 x 1 x 1 up to rows of
10 000 cells, plane sizes that are and are not multiples of 4 or of the tile sizes) and random options:
  cdfpsi   (U, -V, -mean, -full, -mask -ref)            bit for bit
  cdfvT    (r8 / r4 out, -vvl, 1..4 frames)             bit for bit
  cdfcurl  (E-W wrap, -T, -overf, r4 out)               bit for bit
  cdfmhst  (-vt, V/T/S, -Zdim, -full, 1..8 masks)       1e-13 of the integral over |terms|
  cdfheatc (r8 sums; byte / r4 mask, -full, -mxloption) 1e-13
  cdfmean / cdfsum (1..4 basins, three forms, -full)     1e-13
  cdfzonalsum / cdfzonalmean (class / fast / general kernels; extrema bit for bit)   1e-13
  cdfvtrp / cdfvhst vertical integrals                   bit for bit
  cdfmocsig (three equations of state, -full, -eiv, -isodep, class / general kernel)   1e-13
  sharded plans of cdfmoc / cdfpsi / cdfmhst / cdfvT / cdfheatc (reference-order sums), 2..6 row blocks   the bits of one plan
tests/ run fixed shapes; this draws new ones for FUZZ_SECONDS (default 300) and stops at the first failure.
Run on the GPU box:  FUZZ_SECONDS=300 [FUZZ_ONLY=mocsig] python tools/fuzz_tools.py [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_tools  # noqa: E402
import synth  # noqa: E402
from cdftools_amd import hip  # noqa: E402
from test_stencil import frames, oracle_vt_all  # noqa: E402
from test_transport import mhst_masks, tracer_fields  # noqa: E402

f32 = np.float32
seed = int(sys.argv[1]) if len(sys.argv) > 1 else int(time.time()) & 0xffffff
rng = np.random.default_rng(seed)
stop = time.time() + float(os.environ.get("FUZZ_SECONDS", "300"))
LENGTHS = [1, 2, 3, 4, 5, 7, 63, 64, 65, 127, 128, 129, 255, 256, 257, 258, 511, 513, 1023, 1024, 1025, 1442, 4322, 9999]
count = {}


def fail(tool, what, **kw):
    print(f"FAIL seed={seed} tool={tool} {what} {kw}", flush=True)
    sys.exit(1)


while time.time() < stop:
    npi = int(rng.choice(LENGTHS)) if rng.random() < 0.6 else int(rng.integers(1, 3000))
    npj = int(rng.choice([1, 2, 3, 5, 9, 33, 129, 300])) if rng.random() < 0.7 else int(rng.integers(1, 200))
    npk = int(rng.integers(1, 6))
    while npi * npj * npk > 1_500_000:
        npj = max(1, npj // 2)
    tool = ["psi", "vt", "curl", "mhst", "heatc", "mean", "zonal", "vint", "mocsig", "sharded"][int(rng.integers(10))]
    if os.environ.get("FUZZ_ONLY"):                        # FUZZ_ONLY=mocsig: every case of one tool
        tool = os.environ["FUZZ_ONLY"]
    if tool == "mocsig":
        npk = max(npk, 2)
    g = synth.make_grid(npi, npj, max(npk, 2) if tool == "curl" else npk, seed=int(rng.integers(1 << 30)), land=float(rng.choice([0.0, 0.35, 0.8])))
    shp = dict(npi=npi, npj=npj, npk=npk)
    count[tool] = count.get(tool, 0) + 1
    if tool == "psi":
        _, _, u, v = tracer_fields(g, int(rng.integers(1 << 20)))
        kws = [dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"]), dict(zv=v, e1v=g["e1v"], e3v=g["e3v_0"]),
               dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], zv=v, e1v=g["e1v"], e3v=g["e3v_0"]), dict(zu=u, e2u=g["e2u"], e31d=g["e3t_1d"]),
               dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], fmask=g["fmask"][0].astype(f32), iiref=int(rng.integers(1, npi + 1)),
                    ijref=int(rng.integers(1, npj + 1)))]
        kw = kws[int(rng.integers(len(kws)))]
        os.environ["CDFT_PSI_SCAN_WAVE"] = str(int(rng.integers(2)))
        plan = hip.PsiPlan(npi, npj, npk)
        p4, d8 = plan.compute(**kw)
        r4, r8 = oracle_tools.psi(npk, **kw)
        if not (np.array_equal(d8, r8) and np.array_equal(p4, r4)):
            fail(tool, sorted(kw), **shp)
        plan.close()
    elif tool == "vt":
        nfr = int(rng.integers(1, 5))
        u, v, t, s, e3u, e3v = frames(g, nfr, seed=int(rng.integers(1 << 20)))
        vvl, r4 = bool(rng.integers(2)), bool(rng.integers(2))
        ref = oracle_vt_all(u, v, t, s, *((e3u, e3v) if vvl else ()))
        got = hip.vt_compute(u, v, t, s, *((e3u, e3v) if vvl else ()), out_r4=r4)
        for k in ref:
            if not np.array_equal(got[k], ref[k].astype(f32) if r4 else ref[k], equal_nan=True):
                fail(tool, k, nfr=nfr, vvl=vvl, r4=r4, **shp)
    elif tool == "curl":
        nlev = npk
        u, v, _, _, _, _ = frames(g, 1, seed=int(rng.integers(1 << 20)))
        un, vn = u[0, :nlev], v[0, :nlev]
        ff = (2 * (2 * np.pi / 86400.0) * np.sin(g["gphif"].astype(np.float64) * np.pi / 180.0))
        ff[0, :] = 0.0
        lperio, ltpoint, use_ff, r8 = (bool(rng.integers(2)) for _ in range(4))
        os.environ["CDFT_CURL_ROWTILE"] = str(int(rng.integers(2)))
        plan = hip.CurlPlan(npi, npj)
        plan.set_mesh(g["e1u"], g["e2v"], g["e1f"], g["e2f"], ff=ff if use_ff else None, lperio=lperio)
        got = plan.compute(un, vn, ltpoint=ltpoint, out_r8=r8)
        ref = np.stack([oracle_tools.curl(un[k], vn[k], g["e1u"], g["e2v"], g["e1f"], g["e2f"], lperio, ltpoint, ff if use_ff else None)
                        for k in range(nlev)])
        if not np.array_equal(got, ref if r8 else ref.astype(f32)):
            fail(tool, "", lperio=lperio, ltpoint=ltpoint, use_ff=use_ff, r8=r8, **shp)
        plan.close()
    elif tool == "mhst":
        nb = int(rng.choice([1, 4, 4, 8]))
        t, s, _, v = tracer_fields(g, int(rng.integers(1 << 20)))
        m = mhst_masks(g, nb)
        if nb == 8:
            m[4:] = (rng.random((4,) + m.shape[1:]) < 0.5).astype(f32)
        full, zdim, sep = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
        e3 = dict(e31d=g["e3t_1d"]) if full else dict(e3v=g["e3v_0"])
        if sep:
            zdim = False
            fld = dict(zv=v, zt=t, zs=s)
        else:
            vt = ((f32(0.5) * (t + np.roll(t, -1, axis=1))).astype(f32) * v).astype(f32)
            vs = ((f32(0.5) * (s + np.roll(s, -1, axis=1))).astype(f32) * v).astype(f32)
            fld = dict(vt=vt, vs=vs)
        plan = hip.MhstPlan(npi, npj, npk, nb)
        plan.set_mesh(g["e1v"], m, **e3)
        got = plan.compute(zdim=zdim, **fld)
        ref = oracle_tools.mhst(g["e1v"], m, zdim=zdim, **e3, **fld)
        bound = oracle_tools.mhst(g["e1v"], m, zdim=zdim, **e3, **{k: np.abs(x) for k, x in fld.items()})
        for q in (0, 1):
            if not np.all(np.abs(got[q] - ref[q]) <= 1e-13 * bound[q] + 1e-300):
                fail(tool, q, nb=nb, full=full, zdim=zdim, sep=sep, **shp)
        plan.close()
    elif tool == "mean":                                   # cdfmean / cdfsum (src/cdfmean.f90:405-431, src/cdfsum.f90:303-327)
        from test_cli_next import mean_inputs
        nb = int(rng.choice([1, 2, 3, 4]))
        g, zv, ib = mean_inputs(npi, npj, npk, nb, seed=int(rng.integers(1 << 20)))
        tm = g["tmask"].astype(f32)
        form, full = int(rng.integers(4)), bool(rng.integers(2))      # 3 = cdfmean without the variance sum
        e3 = dict(e31d=g["e3t_1d"]) if full else dict(e3=g["e3t_0"])
        plan = hip.MeanPlan(npi, npj, npk, nb)
        plan.set_mesh(g["e1t"], g["e2t"], tm, ib, **e3)
        plan.set_form(form)
        zc = np.where(tm == 0, f32(0), zv).astype(f32) if form in (1, 2) else zv      # (the 1e20 fill of land survives only cdfmean's zv*zmask)
        got = plan.compute(zc)
        ref = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, zc, form=form % 3, **e3)
        bound = oracle_tools.mean(g["e1t"], g["e2t"], tm, ib, np.abs(zc), form=form % 3, **e3)
        if form == 3:
            ref[..., 3] = 0.0
        if not np.all(np.abs(got - ref) <= 1e-13 * np.abs(bound) + 1e-300):
            fail(tool, "", nb=nb, form=form, full=full, **shp)
        plan.close()
    elif tool == "zonal":                                  # cdfzonalsum / cdfzonalmean
        from test_cli_next import zonal_inputs
        nb = int(rng.choice([1, 3, 5]))
        g, zv, zm = zonal_inputs(npi, npj, npk, nb, seed=int(rng.integers(1 << 20)))
        tm = g["tmask"].astype(f32)
        fam = int(rng.integers(3))
        for key in ("CDFT_ZONAL_GENERAL", "CDFT_ZONAL_NOCLASS"):
            os.environ.pop(key, None)
        if fam:
            os.environ["CDFT_ZONAL_GENERAL" if fam == 2 else "CDFT_ZONAL_NOCLASS"] = "1"
        mean_form = bool(rng.integers(2))
        plan = hip.ZonalPlan(npi, npj, npk, nb)
        plan.set_mesh(g["e1t"], g["e2t"], tm, zm)
        got = plan.compute(zv, mean_form=mean_form, minmax=mean_form)
        ref = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, zv, mean_form=mean_form)
        bound = oracle_tools.zonal(g["e1t"], g["e2t"], tm, zm, np.abs(zv), mean_form=mean_form)["sum"]
        ok = np.all(np.abs(got["sum"] - ref["sum"]) <= 1e-13 * bound + 1e-300)
        if mean_form:
            ok = ok and np.all(np.abs(got["area"] - ref["area"]) <= 1e-13 * np.abs(ref["area"]) + 1e-300)
            ok = ok and np.array_equal(got["max"], ref["max"]) and np.array_equal(got["min"], ref["min"])
        if not ok:
            fail(tool, "", nb=nb, family=fam, mean_form=mean_form, **shp)
        plan.close()
    elif tool == "sharded":                                # row-block sharded plans (every block on this device) == one plan
        import oracle_lib
        nd = int(rng.integers(2, 7))
        if npj < nd:
            npj = nd + int(rng.integers(0, 4))
            g = synth.make_grid(npi, npj, npk, seed=int(rng.integers(1 << 30)))
            shp = dict(npi=npi, npj=npj, npk=npk)
        dev = [0] * nd
        t, s, u, v = tracer_fields(g, int(rng.integers(1 << 20)))
        which = ["moc", "psi", "mhst", "vt", "heatc_r4seq"][int(rng.integers(5))]
        ok = True
        if which == "moc" and npk >= 2:
            nb = int(rng.choice([1, 5]))
            e1v, e3v, ib = synth.moc_inputs(g, nbasins=nb)
            zv = synth.velocity(g, int(rng.integers(1, 4)), seed=int(rng.integers(1 << 20)))
            one, sh = hip.MocPlan(npi, npj, npk, nb), hip.MocPlan(npi, npj, npk, nb, devices=dev)
            for pl in (one, sh):
                pl.set_mesh(e1v, e3v, ib)
            ok = np.array_equal(sh.compute(zv), one.compute(zv))
        elif which == "psi":
            kw = [dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"]), dict(zv=v, e1v=g["e1v"], e3v=g["e3v_0"]),
                  dict(zu=u, e2u=g["e2u"], e3u=g["e3u_0"], zv=v, e1v=g["e1v"], e3v=g["e3v_0"])][int(rng.integers(3))]
            kw.update(iiref=int(rng.integers(1, npi + 1)), ijref=int(rng.integers(1, npj + 1)))
            a, b = hip.PsiPlan(npi, npj, npk).compute(**kw), hip.PsiPlan(npi, npj, npk, devices=dev).compute(**kw)
            ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        elif which == "mhst":
            m = mhst_masks(g, 4)
            vt_, vs_ = (t * v).astype(f32), (s * v).astype(f32)
            one, sh = hip.MhstPlan(npi, npj, npk, 4), hip.MhstPlan(npi, npj, npk, 4, devices=dev)
            for pl in (one, sh):
                pl.set_mesh(g["e1v"], m, e3v=g["e3v_0"])
            fld = dict(zv=v, zt=t, zs=s) if rng.integers(2) else dict(vt=vt_, vs=vs_)
            a, b = one.compute(**fld), sh.compute(**fld)
            ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        elif which == "vt":
            fr = frames(g, int(rng.integers(1, 4)), seed=int(rng.integers(1 << 20)))
            a, b = hip.vt_compute(*fr[:4]), hip.vt_compute(*fr[:4], devices=dev)
            ok = all(np.array_equal(a[k], b[k]) for k in a)
        elif which == "heatc_r4seq":                       # the chains carried from block to block: the same bits
            temp = (-2 + 32 * rng.random((npk, npj, npi))).astype(f32)
            one, sh = hip.HeatcPlan(npi, npj, npk), hip.HeatcPlan(npi, npj, npk, devices=dev)
            for pl in (one, sh):
                pl.set_mesh(g["e1t"], g["e2t"], g["tmask"].astype(np.uint8), e3t=g["e3t_0"])
                pl.set_sum_mode(True)
            ok = np.array_equal(one.compute(temp).view(np.uint64), sh.compute(temp).view(np.uint64))
        if not ok:
            fail(tool, which, blocks=nd, **shp)
    elif tool == "mocsig":                                 # cdfmocsig (src/cdfmocsig.f90:368-470): class and general kernels
        import oracle_lib
        nb = int(rng.choice([1, 3, 5]))
        t, s, _, v = tracer_fields(g, int(rng.integers(1 << 20)))
        if rng.integers(2):                                # smooth fields: neighbours share a density class
            t = (t * 0 + (20 - 18 * np.linspace(0, 1, npj, dtype=f32))[None, :, None]).astype(f32) * g["tmask"]
        e1v, e3v, ib = synth.moc_inputs(g, nbasins=nb)
        eos, pref = int(rng.integers(3)), float(rng.choice([0.0, 1000.0, 2000.0]))
        sigmin, sigstp, nbins = {0.0: (20.0, 0.1, 100), 1000.0: (24.0, 0.1, 101), 2000.0: (30.0, 0.05, 158)}[pref]
        if eos == 2:
            pref = 0.0
        full, eiv, isodep = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
        if rng.integers(4) == 0:
            os.environ["CDFT_MOCSIG_NOCLASS"] = "1"
        else:
            os.environ.pop("CDFT_MOCSIG_NOCLASS", None)
        kw = dict(pref=pref, zveiv=(v * f32(0.1)).astype(f32) if eiv else None, gdep=g["gdept_1d"] if isodep else None, isodep=isodep,
                  **(dict(e31d=g["e3t_1d"]) if full else dict(e3v=e3v)))
        got = hip.mocsig(v, t, s, e1v, ib, nbins, sigmin, sigstp, eos=(hip.EOS_80, hip.EOS_TEOS10, hip.EOS_NTR)[eos], **kw)
        ref = oracle_lib.mocsig(v, t, s, e1v, ib, nbins, sigmin, sigstp, teos10=eos == 1, ntr=eos == 2, **kw)
        kwb = dict(kw, zveiv=None if not eiv else np.abs(kw["zveiv"]))
        bound = oracle_lib.mocsig(np.abs(v), t, s, e1v, ib, nbins, sigmin, sigstp, teos10=eos == 1, ntr=eos == 2, **kwb)
        gd, rd = (got[0], ref[0]) if isodep else (got, ref)
        bd = bound[0] if isodep else bound
        ok = np.all(np.abs(gd - rd) <= 1e-13 * np.abs(bd) + 1e-300)
        if isodep:
            ok = ok and np.allclose(got[1], ref[1], rtol=1e-12, atol=0)
        if not ok:
            fail(tool, "", nb=nb, eos=eos, pref=pref, full=full, eiv=eiv, isodep=isodep, noclass=os.environ.get("CDFT_MOCSIG_NOCLASS"), **shp)
    elif tool == "vint":                                   # cdfvtrp / cdfvhst: every cell walks its levels in order: bit for bit
        t, s, u, v = tracer_fields(g, int(rng.integers(1 << 20)))
        full = bool(rng.integers(2))
        e3 = dict(e31d=g["e3t_1d"]) if full else dict(e3=g["e3v_0"])
        if rng.integers(2):
            got, ref = hip.vint_transport(v, g["e1v"], **e3), oracle_tools.vint_transport(v, g["e1v"], **e3)
            ok = np.array_equal(got, ref)
        else:
            vt, vs = (t * v).astype(f32), (s * v).astype(f32)
            got, ref = hip.vhst_transport(vt, vs, g["e1v"], **e3), oracle_tools.vhst_transport(vt, vs, g["e1v"], **e3)
            ok = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        if not ok:
            fail(tool, "", full=full, **shp)
    else:
        shape = (npk, npj, npi)
        e3t = (1 + 199 * rng.random(shape)).astype(f32)
        e31d = (1 + 199 * rng.random(npk)).astype(f32)
        mask = (rng.random(shape) > 0.4).astype(f32)
        temp = (-2 + 32 * rng.random(shape)).astype(f32)
        mxl = (10 + 300 * rng.random((npj, npi))).astype(f32)
        gdepw = np.cumsum(np.concatenate([[0], e31d[:-1]])).astype(f32)
        full, m8, opt = bool(rng.integers(2)), bool(rng.integers(2)), int(rng.integers(-1, 2))
        m = mask.astype(np.uint8) if m8 else (mask * f32(0.75)).astype(f32)
        e3 = dict(e31d=e31d) if full else dict(e3t=e3t)
        plan = hip.HeatcPlan(npi, npj, npk)
        plan.set_mesh(g["e1t"], g["e2t"], m, **e3)
        plan.set_sum_mode(False)
        got = plan.compute(temp, rmxldep=mxl if opt else None, gdepw=gdepw, mxloption=opt)
        kw = dict(mxl=mxl if opt else None, gdepw=gdepw, mxlopt=opt)
        ref = oracle_tools.heatc(g["e1t"], g["e2t"], m.astype(f32), temp, **e3, **kw)
        bound = oracle_tools.heatc(g["e1t"], g["e2t"], np.abs(m.astype(f32)), np.abs(temp), **e3, **kw)
        if not np.all(np.abs(got - ref) <= 1e-13 * np.abs(bound) + 1e-300):
            fail(tool, "", full=full, m8=m8, opt=opt, **shp)
        plan.close()
print(f"fuzz_tools: seed {seed}: {sum(count.values())} cases {count}: every result equal to the oracle (psi, vt, curl, vint: bit for bit; mhst, heatc, mean, zonal, mocsig: 1e-13)")
