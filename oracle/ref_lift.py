#!/usr/bin/env python3
"""oracle/ref_lift.py -- harnesses around LITERAL statements of the reference programs (TEST INFRASTRUCTURE).

What this is.  The cdf*.f90 programs of the reference cannot be built in this image: they reach NetCDF
through cdfio.F90 and netcdf-fortran is absent.  Their ARITHMETIC, however, is plain
Fortran.  This script reads /root/reference/src/<tool>.f90 where it lies and writes, into the git-ignored
oracle/_ref/, one Fortran subroutine per tool made of

  * the program's own declaration block, lifted verbatim (only `TYPE(variable)` lines, which need cdfio, are
    dropped), so every variable has the reference's type, kind and rank;
  * the program's own ALLOCATE statements and its whole time loop, lifted verbatim by line range;
  * harness glue written here: a BIND(C) argument list of plain arrays, flag assignments that stand for the
    command-line parser, and captures of results.

The only lines of a lifted range that are altered are I/O statements, and only in these mechanical ways:
  `getvar(file, var, lev, npiglo, npjglo [, ktime=t] ...)`  ->  `h_<array>(:,:,lev,t)`  (a plain array argument)
  `getvare3(file, var, n)` / `getvar1d(file, var, n)`        ->  `h_<array>(:)`
  `ierr = putvar(ncout, id_varout(n), EXPR, lev, ...)`       ->  `o_<array>(..) = EXPR`  (EXPR untouched)
  `PRINT ...` / `WRITE(...)`                                  ->  dropped (or, where the tool spec says so, a capture of
                                                                 the printed variables)
  `ierr = closeout(..)`, `STOP n`                             ->  `ierr = 0`, `RETURN`
Every other line reaches the compiler byte for byte; `lift()` asserts that a line it alters holds one of those
I/O tokens, and each lifted range is pinned by the SHA-256 of its text so that a different reference revision is
refused instead of silently shifting line numbers.

What this is NOT.  It is not the reference program and not a build of it: no cdfio, no files, no option parser.
It pins the evaluation order, the r4/r8 mixing and the index ranges of the loops -- the part a transcription can
get wrong -- because the compiler sees the reference's own text.  The generated sources and the shared object
stay in oracle/_ref/ (git-ignored): reference text never enters the repository.  Only outputs of the harness run
on synthetic inputs are committed (tests/golden/lift_*.npz, made by tests/make_golden_lift.py).

Usage: python3 oracle/ref_lift.py [--out DIR] [--shas]     (called by oracle/ref_build.sh; needs /root/reference)
"""
import hashlib
import os
import re
import sys

REF = os.environ.get("CDFT_REFERENCE_SRC", "/root/reference/src")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

IO_TOKEN = re.compile(r"\b(getvar|getvare3|getvar1d|putvar|putvar1d|putvar0d|closeout|PRINT|WRITE|STOP|OPEN|CLOSE)\b", re.I)


# SHA-256 (first 16 hex digits) of every lifted range at the reference revision this was written for
# (`python3 oracle/ref_lift.py --shas` prints the current ones).
SHAS = {
    "cdfmoc.f90": {'49-125': '3fefe705db05454d', '236-236': '75b30441aecc08bb', '272-278': '105e320aa63f3b97',
                   '281-304': 'e8ff1410b994b5fb', '306-314': '5c9fb47f8397cf47', '325-336': 'b805d15787c1cb5d',
                   '338-518': '93b908f104591dd7', '519-570': '30dc9880c21c4367', '576-596': '4dddff438b339b54',
                   '618-653': '254e9a3188c2bfa2', '683-690': 'be10b9f6d92d9d34', '692-705': 'bbcc7224fd6d58c3',
                   '709-998': '7b7fa589a3f32135'},
    "cdfheatc.f90": {'24-67': 'f588c6224654319f', '177-186': '4e0de2a06c88c8d6', '188-192': 'b770d4ce667f8e40',
                     '196-248': '1c9fc2a727732949'},
    "cdfmhst.f90": {'34-103': '7ef454039bb47b19', '256-274': 'c4130977d91a663b', '276-277': 'f20e8805749476d9',
                    '285-285': '0c09e2d78019206d', '302-446': 'dd7740ee803f57b3'},
    "cdfpsi.f90": {'34-88': '59db5ae980e25bce', '228-239': 'c25b3e7e171a1692', '246-253': '3cdebe7afb78924e',
                   '256-423': 'faea68189a77a289', '424-439': '54abc3e06a253c80'},
    "cdfvT.f90": {'29-66': '01a91a9f11bdef51', '146-150': 'be34dddcb49a65d7', '152-155': '1ceb1ad495d85ecd',
                  '159-159': '355959c032428590', '160-168': 'e7924de889e9fd84', '186-259': '190b74a4951ddade'},
    "cdfcurl.f90": {'26-64': '2e0ba28c42a38070', '190-198': '1cf01aed8a82cb4b', '200-225': 'b703bc172976af47',
                    '238-238': '1e01b69985fc68f0', '242-304': '436d55fbb38c6d62'},
    "cdfvhst.f90": {'25-56': '3fabc6e253b698e3', '129-135': 'b79d5deead59e1a4', '140-142': '09aecf481b63a53f',
                    '144-181': 'dc19234ad642a260'},
    "cdfvtrp.f90": {'27-62': '3a1911d6740770e8', '151-163': 'abe836d83ab5b2a5', '167-176': '2982ef76148e7ef4',
                    '178-252': 'aa4ef0b140e159e0'},
    "cdfsum.f90": {'29-79': '70fa8b3dc7293818', '214-217': 'a237d141ae03476c', '275-286': '9eef53076f4ffe91',
                   '290-349': 'c55d0345bfb35f54'},
    "cdfmean.f90": {'27-109': '2afc692c05a8c49c', '294-303': '3810e05a885119c3', '361-366': '30c75f9da2454881',
                    '368-373': '2d3f4324de749945', '376-378': '16a1fb636762cdc2', '383-389': '25303bff04b9110c',
                    '396-482': 'd620d6b665ebeee5'},
    "cdfvertmean.f90": {'28-79': 'c8171621ff04d256', '183-188': '0df32f3e32228af5', '210-218': '733b29a12144d7b3',
                        '224-269': '168e09af6bf3b12c'},
    "cdfvint.f90": {'28-68': 'bd62e334b804c648', '234-242': 'dc6d815578f78f00', '245-259': '8cc915212078de44',
                    '268-313': 'b699a0f2a22070ff'},
    "cdfzonalsum.f90": {'27-83': 'ad0b9842a9a961ac', '246-252': '0882565e17aedd7e', '255-280': '6fa18a4d6c72f75a',
                        '286-294': 'dcf9e73016b19a2c', '297-326': '9bd815763a8a2722'},
    "cdfzonalmean.f90": {'27-84': 'a13dce4e4fd7aad5', '248-256': '1a4ec079c4829d95', '259-278': '150e86ef5d5a1311',
                         '287-297': '73cf4242f36fa74f', '300-350': '2349b9d6ecf65284'},
    "cdfmocsig.f90": {'40-108': '9b8d04603f1640c1', '255-261': '37870c2cf98ed885', '299-304': '24fccd9c0d50516c',
                      '308-323': 'd5334578d236b66e', '325-328': 'ab5348a0318ad2fd', '342-342': 'bc3b9905f14f22c2',
                      '347-359': 'bda9394497041e7f', '361-476': '2a40e249c283bba6', '477-485': '1a302182f99b7d53'},
    "cdfmaxmoc.f90": {'28-59': 'b55b09bb29db5292', '122-122': '8aa24a0caec32bf0', '123-124': '347ce8e20ebb08e9',
                      '140-178': '195ad4300d453a07'},
    "cdfsig0.f90": {'25-51': '70cc5ee89f5abbd9', '132-134': 'd7f186c3f2f1c40e', '139-155': 'e6971df41b8c61f4'},
    "cdfsigi.f90": {'26-52': '9e777513b6fb83e8', '124-126': '6a9aca4456a24830', '130-145': '40727884d7f41266'},
}


def split_args(s):
    """Split a Fortran argument list at top-level commas."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def find_call(line, name):
    """-> (start, end, args) of the first `name ( ... )` call in line, or None."""
    m = re.search(r"\b%s\s*\(" % name, line)
    if not m:
        return None
    i = m.end()
    depth = 1
    while i < len(line) and depth:
        if line[i] == "(":
            depth += 1
        elif line[i] == ")":
            depth -= 1
        i += 1
    if depth:
        raise ValueError("unbalanced call in: " + line)
    return m.start(), i, split_args(line[m.end():i - 1])


def varidx(tok):
    """`id_varout(n)` -> `n`"""
    return re.sub(r"id_varout\s*\((.*)\)", r"\1", tok)


class Tool:
    def __init__(self, fname, varmap, putvar=None, capture=None, shas=None):
        self.fname = fname
        with open(os.path.join(REF, fname)) as f:
            self.src = f.read().split("\n")
        self.varmap = varmap            # reference variable-name token -> harness array
        self.putvar = putvar            # callable(args, kw) -> statement text
        self.capture = capture or {}    # line number -> replacement for a PRINT / WRITE line
        self.shas = shas if shas is not None else SHAS.get(fname, {})
        self.report = []

    def text(self, a, b):
        return [ln.rstrip() for ln in self.src[a - 1:b]]

    def check_sha(self, a, b):
        h = hashlib.sha256("\n".join(self.text(a, b)).encode()).hexdigest()[:16]
        key = f"{a}-{b}"
        want = self.shas.get(key)
        self.report.append((key, h))
        if want is None and "--shas" not in sys.argv:
            raise SystemExit(f"{self.fname}:{key}: range has no pinned SHA")
        if want is not None and want != h:
            raise SystemExit(f"{self.fname}:{key}: reference text differs from the revision this harness was written "
                             f"for (sha {h}, expected {want})")

    def decls(self, a, b):
        """The declaration block, verbatim, minus TYPE(variable) lines (need cdfio)."""
        self.check_sha(a, b)
        out = []
        for ln in self.text(a, b):
            if re.match(r"\s*TYPE\s*\(\s*variable\s*\)", ln, re.I):
                out.append("  ! [lift] dropped: " + ln.strip())
            else:
                out.append(ln)
        return out

    def raw(self, a, b):
        """Lines a..b verbatim; refuses ranges that hold I/O."""
        self.check_sha(a, b)
        for n, ln in enumerate(self.text(a, b), a):
            code = ln.split("!")[0]
            if IO_TOKEN.search(code):
                raise SystemExit(f"{self.fname}:{n}: raw() range holds an I/O statement: {ln}")
        return self.text(a, b)

    def _kw(self, args):
        pos, kw = [], {}
        for a in args:
            m = re.match(r"^(\w+)\s*=\s*(.*)$", a)
            if m and not a.startswith(("'", '"')):
                kw[m.group(1).lower()] = m.group(2)
            else:
                pos.append(a)
        return pos, kw

    def rewrite(self, n, ln):
        code = ln.split("!")[0] if "'" not in ln and '"' not in ln else ln
        if not IO_TOKEN.search(code):
            return ln                                           # arithmetic / control flow: byte for byte
        if n in self.capture:
            return self.capture[n]
        s = ln
        if re.match(r"\s*(PRINT|WRITE|OPEN|CLOSE)\b", s, re.I) or re.search(r"\)\s*(PRINT|WRITE|OPEN|CLOSE)\b", code, re.I):
            return "  ! [lift] I/O dropped: " + s.strip()
        for name in ("getvare3", "getvar1d"):
            c = find_call(s, name)
            if c:
                pos, kw = self._kw(c[2])
                arr = self.varmap[pos[1]]
                return s[:c[0]] + f"{arr}(:)" + s[c[1]:]
        c = find_call(s, "getvar")
        if c:
            pos, kw = self._kw(c[2])
            arr = self.varmap[pos[1]]
            lev = pos[2]
            t = kw.get("ktime", "1")
            if "(" in arr:                                      # the tool spec gives the whole array section
                return s[:c[0]] + arr.format(lev=lev, t=t) + s[c[1]:]
            return s[:c[0]] + f"{arr}(:,:,{lev},{t})" + s[c[1]:]
        c = find_call(s, "putvar1d")
        if c:
            m = re.search(r"\bierr\s*=\s*$", s[:c[0]])
            return s[:m.start()] + "ierr = 0" + s[c[1]:]
        c = find_call(s, "putvar0d")
        if c:
            pos, kw = self._kw(c[2])
            m = re.search(r"\bierr\s*=\s*$", s[:c[0]])
            return s[:m.start()] + self.putvar0d(pos, kw) + s[c[1]:]
        c = find_call(s, "putvar")
        if c:
            pos, kw = self._kw(c[2])
            m = re.search(r"\bierr\s*=\s*$", s[:c[0]])
            if not m:
                raise SystemExit(f"{self.fname}:{n}: putvar not of the form `ierr = putvar(...)`: {ln}")
            return s[:m.start()] + self.putvar(pos, kw) + s[c[1]:]
        c = find_call(s, "closeout")
        if c:
            m = re.search(r"\bierr\s*=\s*$", s[:c[0]])
            return s[:m.start()] + "ierr = 0" + s[c[1]:]
        m = re.search(r"\bSTOP\b\s*\d*", s)
        if m:
            return s[:m.start()] + "RETURN" + s[m.end():]
        raise SystemExit(f"{self.fname}:{n}: I/O token without a rewrite rule: {ln}")

    def lift(self, a, b):
        self.check_sha(a, b)
        out, lines, i = [], self.text(a, b), 0
        while i < len(lines):
            n, ln = a + i, lines[i]
            code = ln.split("!")[0]
            # an I/O call continued over several lines (`getvar(... , &` / `& ktime=it ...)`) is joined before it is rewritten
            while IO_TOKEN.search(code) and code.rstrip().endswith("&") and i + 1 < len(lines):
                i += 1
                nxt = lines[i].split("!")[0].strip()
                ln = code.rstrip()[:-1].rstrip() + " " + (nxt[1:].lstrip() if nxt.startswith("&") else nxt)
                code = ln
            out.append(self.rewrite(n, ln))
            i += 1
        return out

    def expand(self, template):
        out = []
        for ln in template.split("\n"):
            m = re.match(r"\s*!@(DECLS|LIFT|RAW)\s+(\d+)\s+(\d+)", ln)
            if not m:
                out.append(ln)
                continue
            a, b = int(m.group(2)), int(m.group(3))
            out.append(f"  ! ---- {m.group(1).lower()} {self.fname}:{a}-{b}")
            out += {"DECLS": self.decls, "LIFT": self.lift, "RAW": self.raw}[m.group(1)](a, b)
            out.append(f"  ! ---- end {self.fname}:{a}-{b}")
        return "\n".join(out) + "\n"


# =============================================================================================== cdfmoc
# src/cdfmoc.f90: declarations :49-125, eos_init :236, sizes :272-278, ALLOCATE :281-304, metrics :306-314,
# masks :325-336, the whole time loop :338-570 (zonal sums, cumulated sums, -decomp, output), get_e3v :576-596.
CDFMOC = r"""
SUBROUTINE lift_cdfmoc(kpi, kpj, kpk, kpt, kbas, kdec, kfull, kteos10, knvar, kvvl, h_e1v, h_e1u, h_gphiv, h_gdepw, h_gdept, &
     &  h_e31d, h_vmask, h_umask, h_tmask, h_e3v, h_atl, h_ind, h_pac, h_v, h_t, h_s, o_var, o_dmoc, o_dec) &
     &  BIND(C, name='lift_cdfmoc')
  USE ISO_C_BINDING
  USE eos                      ! the reference's src/eos.f90, compiled where it lies (ref_build.sh)
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kbas, kdec, kfull, kteos10, knvar, kvvl   ! kvvl 0 / 1: -vvl, e3v of every record
  REAL(C_FLOAT), INTENT(in) :: h_e1v(kpi,kpj,1,1), h_e1u(kpi,kpj,1,1), h_gphiv(kpi,kpj,1,1)
  REAL(C_FLOAT), INTENT(in) :: h_gdepw(kpk), h_gdept(kpk), h_e31d(kpk)
  REAL(C_FLOAT), INTENT(in) :: h_vmask(kpi,kpj,kpk,1), h_umask(kpi,kpj,kpk,1), h_tmask(kpi,kpj,kpk,1), h_e3v(kpi,kpj,kpk,1+(kpt-1)*kvvl)
  REAL(C_FLOAT), INTENT(in) :: h_atl(kpi,kpj,1,1), h_ind(kpi,kpj,1,1), h_pac(kpi,kpj,1,1)
  REAL(C_FLOAT), INTENT(in) :: h_v(kpi,kpj,kpk,kpt), h_t(kpi,kpj,kpk,kpt), h_s(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT),  INTENT(out) :: o_var(kpj,kpk,knvar,kpt)       ! what putvar is handed: REAL(...) rows
  REAL(C_DOUBLE), INTENT(out) :: o_dmoc(kbas,kpj,kpk,kpt)       ! dmoc after the cumulated sum
  REAL(C_DOUBLE), INTENT(out) :: o_dec(kbas,kpj,kpk,3,kpt)      ! dmoc_sh, dmoc_bt, dmoc_ag (-decomp)
  LOGICAL :: lg_vvl = .FALSE.      ! modcdfnames variable; -vvl changes the file e3v is read from and reads it at every record
!@DECLS 49 125
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt
  lbas = ( kbas == 5 ) ; ldec = ( kdec /= 0 ) ; lfull = ( kfull /= 0 ) ; ll_teos10 = ( kteos10 /= 0 )
  lg_vvl = ( kvvl /= 0 )
!@RAW 236 236
!@RAW 272 278
  ! basin indices as CreateOutput numbers them (src/cdfmoc.f90:1032,:1057,:1081,:1106,:1131,:1156)
  npglo = 1 ; npatl = 2 ; npinp = 3 ; npind = 4 ; nppac = 5 ; npinp0 = 6
!@RAW 281 304
!@LIFT 306 314
!@LIFT 325 336
  IF ( ldec ) dmoc_sh = 0.d0     ! ALLOCATE leaves it undefined and the reference never clears it (:471); start from 0
!@LIFT 338 518
     o_dmoc(:,:,:,jt) = dmoc(:,:,:)
     IF ( ldec ) THEN
        o_dec(:,:,:,1,jt) = dmoc_sh ; o_dec(:,:,:,2,jt) = dmoc_bt ; o_dec(:,:,:,3,jt) = dmoc_ag
     ENDIF
!@LIFT 519 570
CONTAINS
!@LIFT 576 596
END SUBROUTINE lift_cdfmoc
"""


def gen_cdfmoc():
    t = Tool("cdfmoc.f90",
             {"cn_ve1v": "h_e1v", "cn_gphiv": "h_gphiv", "cn_gdepw": "h_gdepw", "cn_gdept": "h_gdept", "cn_ve1u": "h_e1u",
              "cn_ve3t1d": "h_e31d", "cn_vmask": "h_vmask", "cn_tmaskatl": "h_atl", "cn_tmaskind": "h_ind",
              "cn_tmaskpac": "h_pac", "cn_vomecrty": "h_v", "cn_umask": "h_umask", "cn_tmask": "h_tmask",
              "cn_votemper": "h_t", "cn_vosaline": "h_s", "cn_ve3v": "h_e3v"},
             putvar=lambda pos, kw: f"o_var(:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFMOC)


# =============================================================================================== cdfheatc
# src/cdfheatc.f90: declarations :24-67, ALLOCATE :177-186, metrics :188-192, time loop :196-248.
# The PRINT of :233 / :235 shows dsurf, dsum2d per level: those two lines become the capture of the level sums.
CDFHEATC = r"""
SUBROUTINE lift_cdfheatc(kpi, kpj, kpk, kpt, kfull, kmxl, kvvl, h_e1t, h_e2t, h_gdepw, h_e31d, h_tmask, h_e3t, h_t, h_mxl, &
     &  o_var, o_sums) BIND(C, name='lift_cdfheatc')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kfull, kmxl, kvvl
  REAL(C_FLOAT), INTENT(in) :: h_e1t(kpi,kpj,1,1), h_e2t(kpi,kpj,1,1), h_gdepw(kpk), h_e31d(kpk)
  REAL(C_FLOAT), INTENT(in) :: h_tmask(kpi,kpj,kpk,1), h_e3t(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_t(kpi,kpj,kpk,kpt), h_mxl(kpi,kpj,1,kpt)
  REAL(C_DOUBLE), INTENT(out) :: o_var(kpk,3,kpt)       ! what putvar is handed: heatc3d, heatc2d(k), heatc3dpervol
  REAL(C_DOUBLE), INTENT(out) :: o_sums(3,kpk,kpt)      ! dsurf, dvol2d, dsum2d of each level
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 24 67
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; npkk = kpk ; nvpk = kpk
  iimin = 1 ; ijmin = 1 ; ikmin = 1 ; ikmax = kpk          ! no -zoom: src/cdfheatc.f90:162-164
  lfull = ( kfull /= 0 ) ; mxloption = kmxl ; lg_vvl = ( kvvl /= 0 )
!@RAW 177 186
!@LIFT 188 192
!@LIFT 196 248
END SUBROUTINE lift_cdfheatc
"""


def gen_cdfheatc():
    cap = "           o_sums(1,ik,jt) = dsurf ; o_sums(2,ik,jt) = dvol2d ; o_sums(3,ik,jt) = dsum2d   ! [lift] capture of the PRINT"
    t = Tool("cdfheatc.f90",
             {"cn_ve1t": "h_e1t", "cn_ve2t": "h_e2t", "cn_gdepw": "h_gdepw", "cn_ve3t1d": "h_e31d", "cn_somxl010": "h_mxl",
              "cn_votemper": "h_t", "cv_msk": "h_tmask", "cn_ve3t": "h_e3t"},
             putvar=lambda pos, kw: f"o_var({pos[3]}:{pos[3]},{varidx(pos[1])}:{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}",
             capture={233: cap, 235: cap})
    return t, t.expand(CDFHEATC)


# =============================================================================================== cdfmhst
# src/cdfmhst.f90: declarations :34-103, ALLOCATE :256-274, metrics :276-277,:285, time loop :302-446 (the .dat dumps
# that follow, :449-492, are formatted WRITEs of the same dzonal_* arrays and are not lifted).
CDFMHST = r"""
SUBROUTINE lift_cdfmhst(kpi, kpj, kpk, kpt, kbas, ksepf, kfull, kzdim, kmst, kvvl, h_e1v, h_gphiv, h_gdept, h_e31d, h_e3v, h_vmask, &
     &  h_atl, h_pac, h_ind, h_vt, h_vs, h_v, h_t, h_s, o_var, o_zon) BIND(C, name='lift_cdfmhst')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kbas, ksepf, kfull, kzdim, kmst, kvvl
  REAL(C_FLOAT), INTENT(in) :: h_e1v(kpi,kpj,1,1), h_gphiv(kpi,kpj,1,1), h_gdept(kpk), h_e31d(kpk), h_e3v(kpi,kpj,kpk,1+(kpt-1)*kvvl)
  REAL(C_FLOAT), INTENT(in) :: h_vmask(kpi,kpj,1,1), h_atl(kpi,kpj,1,1), h_pac(kpi,kpj,1,1), h_ind(kpi,kpj,1,1)
  REAL(C_FLOAT), INTENT(in) :: h_vt(kpi,kpj,kpk,kpt), h_vs(kpi,kpj,kpk,kpt), h_v(kpi,kpj,kpk,kpt), h_t(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(in) :: h_s(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT),  INTENT(out) :: o_var(kpj,kpk,12,kpt)   ! REAL(dmtrp) as handed to putvar (level 1 only without -Zdim)
  REAL(C_DOUBLE), INTENT(out) :: o_zon(kpj,8,kpt)        ! dzonal_{heat,salt}_{glo,atl,pac,ind} after the last level
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 34 103
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt
  lsepf = ( ksepf /= 0 ) ; lfull = ( kfull /= 0 ) ; lzdim = ( kzdim /= 0 ) ; lg_vvl = ( kvvl /= 0 )
  llglo = ( kbas == 5 ) ; nbasins = kbas ; nbasinso = 1 ; IF ( llglo ) nbasinso = 6     ! src/cdfmhst.f90:219-226
  npvar = 1 ; IF ( kmst /= 0 ) npvar = 2                                                ! -MST
  npko = 1 ; IF ( lzdim ) npko = npk                                                    ! :247-253
  o_var = 0. ; o_zon = 0.d0
!@RAW 256 274
!@LIFT 276 277
!@LIFT 285 285
!@LIFT 302 446
     o_zon(:,1,jt) = dzonal_heat_glo ; o_zon(:,5,jt) = dzonal_salt_glo
     IF ( llglo ) THEN
        o_zon(:,2,jt) = dzonal_heat_atl ; o_zon(:,3,jt) = dzonal_heat_pac ; o_zon(:,4,jt) = dzonal_heat_ind
        o_zon(:,6,jt) = dzonal_salt_atl ; o_zon(:,7,jt) = dzonal_salt_pac ; o_zon(:,8,jt) = dzonal_salt_ind
     ENDIF
  END DO
END SUBROUTINE lift_cdfmhst
"""


def gen_cdfmhst():
    t = Tool("cdfmhst.f90",
             {"cn_vomecrty": "h_v", "cn_votemper": "h_t", "cn_vosaline": "h_s", "cn_vomevt": "h_vt", "cn_vomevs": "h_vs",
              "cn_ve3v": "h_e3v", "cn_vmask": "h_vmask", "cn_tmaskatl": "h_atl", "cn_tmaskpac": "h_pac", "cn_tmaskind": "h_ind",
              "cn_ve1v": "h_e1v", "cn_gphiv": "h_gphiv", "cn_ve3t1d": "h_e31d", "cn_gdept": "h_gdept"},
             putvar=lambda pos, kw: f"o_var(:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFMHST)


# =============================================================================================== cdfpsi
# src/cdfpsi.f90: declarations :34-88, ALLOCATE :228-239, metrics and mask :246-253, the whole time loop :256-439.
CDFPSI = r"""
SUBROUTINE lift_cdfpsi(kpi, kpj, kpk, kpt, ku, kv, kfull, kmask, kmean, kopen, kssh, klev, kiref, kjref, h_e1v, h_e2u, h_fmask, &
     &  h_e31d, h_e3u, h_e3v, h_u, h_v, h_ssh, knlo, kvvl, o_var, o_dpsi) BIND(C, name='lift_cdfpsi')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, ku, kv, kfull, kmask, kmean, kopen, kssh, klev, kiref, kjref, knlo, kvvl
  REAL(C_FLOAT), INTENT(in) :: h_e1v(kpi,kpj,1,1), h_e2u(kpi,kpj,1,1), h_fmask(kpi,kpj,1,1), h_e31d(kpk)
  REAL(C_FLOAT), INTENT(in) :: h_e3u(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_e3v(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_u(kpi,kpj,kpk,kpt), h_v(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(in) :: h_ssh(kpi,kpj,1,kpt)
  REAL(C_FLOAT),  INTENT(out) :: o_var(kpi,kpj,knlo,3,kpt)   ! what putvar is handed (SNGL(...) fields)
  REAL(C_DOUBLE), INTENT(out) :: o_dpsi(kpi,kpj,kpt)         ! dpsi after the normalisation, before SNGL
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 34 88
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt
  ll_u = ( ku /= 0 ) ; ll_v = ( kv /= 0 ) ; lfull = ( kfull /= 0 ) ; lmask = ( kmask /= 0 ) ; lmean = ( kmean /= 0 )
  lopen = ( kopen /= 0 ) ; lssh = ( kssh /= 0 ) ; llev = ( klev /= 0 ) ; lg_vvl = ( kvvl /= 0 )
  iiref = kiref ; ijref = kjref
  o_var = 0. ; o_dpsi = 0.d0
!@RAW 228 239
!@LIFT 246 253
!@LIFT 256 423
     o_dpsi(:,:,jt) = dpsi(:,:)
!@LIFT 424 439
END SUBROUTINE lift_cdfpsi
"""


def gen_cdfpsi():
    t = Tool("cdfpsi.f90",
             {"cn_sossheig": "h_ssh", "cn_vomecrty": "h_v", "cn_ve3v": "h_e3v", "cn_vozocrtx": "h_u", "cn_ve3u": "h_e3u",
              "cn_ve1v": "h_e1v", "cn_ve2u": "h_e2u", "cn_fmask": "h_fmask", "cn_ve3t1d": "h_e31d"},
             putvar=lambda pos, kw: f"o_var(:,:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFPSI)


# =============================================================================================== cdfvT
# src/cdfvT.f90: declarations :29-66, ALLOCATE :146-150,:152-155, level loop :159-168 and :186-259.  Lines :169-185 resolve the
# file names of a tag and read its frame count and time axis (SetFileName / getdim / getvar1d): the harness feeds the
# frames of all tags as one array, ntags tags of kfr frames each, so those lines are replaced by `npt = kfr`.
CDFVT = r"""
SUBROUTINE lift_cdfvt(kpi, kpj, kpk, ktags, kfr, kvvl, h_u, h_v, h_t, h_s, h_e3u, h_e3v, o_var) BIND(C, name='lift_cdfvt')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, ktags, kfr, kvvl
  REAL(C_FLOAT), INTENT(in) :: h_u(kpi,kpj,kpk,kfr,ktags), h_v(kpi,kpj,kpk,kfr,ktags), h_t(kpi,kpj,kpk,kfr,ktags)
  REAL(C_FLOAT), INTENT(in) :: h_s(kpi,kpj,kpk,kfr,ktags), h_e3u(kpi,kpj,kpk,kfr,ktags), h_e3v(kpi,kpj,kpk,kfr,ktags)
  REAL(C_DOUBLE), INTENT(out) :: o_var(kpi,kpj,kpk,6)      ! the r8 arrays handed to putvar: vt, vs, ut, us [, e3u, e3v]
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 29 66
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; ntags = ktags ; lg_vvl = ( kvvl /= 0 )
  o_var = 0.d0
  ALLOCATE ( ctag_lst(ntags) )
!@RAW 146 150
!@RAW 152 155
!@RAW 159 159
!@LIFT 160 168
        npt = kfr                                           ! [lift] stands for :169-185 (file names, getdim, time axis)
!@LIFT 186 259
END SUBROUTINE lift_cdfvt
"""


def gen_cdfvt():
    t = Tool("cdfvT.f90", {})
    # getvar(file, var, jk, npiglo, npjglo, ktime=jtt) -> frame jtt of tag jt
    names = {"cn_vozocrtx": "h_u", "cn_vomecrty": "h_v", "cn_votemper": "h_t", "cn_vosaline": "h_s", "cn_ve3u": "h_e3u",
             "cn_ve3v": "h_e3v"}

    class VM(dict):
        def __getitem__(self, k):
            return names[k]
    t.varmap = VM()
    orig = t.rewrite

    def rewrite(n, ln):
        out = orig(n, ln)
        return re.sub(r"(h_\w+)\(:,:,(\w+),(\w+)\)", r"\1(:,:,\2,\3,jt)", out) if out is not ln else out
    t.rewrite = rewrite
    t.putvar = lambda pos, kw: f"o_var(:,:,{pos[3]},{varidx(pos[1])}) = {pos[2]}"
    return t, t.expand(CDFVT)


# =============================================================================================== cdfcurl
# src/cdfcurl.f90: declarations :26-64, ALLOCATE :190-198, metrics and Coriolis :200-225, periodicity :238, time loop
# :242-304.  fmask is only assigned for ji < npiglo, jj < npjglo (:267-273) while -T reads its last row and column
# (:290): the reference leaves them as ALLOCATE found them; the harness clears the array once, as a zeroed heap would.
CDFCURL = r"""
SUBROUTINE lift_cdfcurl(kpi, kpj, kpk, kpt, knlev, kforcing, ktpoint, koverf, h_e1u, h_e1f, h_e2v, h_e2f, h_glamf, h_gphif, &
     &  h_u, h_v, o_var) BIND(C, name='lift_cdfcurl')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, knlev, kforcing, ktpoint, koverf
  REAL(C_FLOAT), INTENT(in) :: h_e1u(kpi,kpj,1,1), h_e1f(kpi,kpj,1,1), h_e2v(kpi,kpj,1,1), h_e2f(kpi,kpj,1,1)
  REAL(C_FLOAT), INTENT(in) :: h_glamf(kpi,kpj,1,1), h_gphif(kpi,kpj,1,1), h_u(kpi,kpj,kpk,kpt), h_v(kpi,kpj,kpk,kpt)
  REAL(C_DOUBLE), INTENT(out) :: o_var(kpi,kpj,knlev,kpt)   ! drotn as handed to putvar
!@DECLS 26 64
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; nlev = knlev
  lforcing = ( kforcing /= 0 ) ; ltpoint = ( ktpoint /= 0 ) ; loverf = ( koverf /= 0 ) ; l_metric = .TRUE. ; lperio = .FALSE.
  ALLOCATE ( nilev(nlev) )
  DO jk = 1, nlev ; nilev(jk) = jk ; ENDDO               ! -l 1 2 ... nlev
!@RAW 190 198
  fmask(:,:) = 0.
!@LIFT 200 225
!@RAW 238 238
!@LIFT 242 304
END SUBROUTINE lift_cdfcurl
"""


def gen_cdfcurl():
    t = Tool("cdfcurl.f90",
             {"cn_ve1u": "h_e1u", "cn_ve1f": "h_e1f", "cn_ve2v": "h_e2v", "cn_ve2f": "h_e2f", "cn_glamf": "h_glamf",
              "cn_gphif": "h_gphif", "cv_u": "h_u", "cv_v": "h_v"},
             putvar=lambda pos, kw: f"o_var(:,:,{pos[3]},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFCURL)


# =============================================================================================== cdfvhst
# src/cdfvhst.f90: declarations :25-56, ALLOCATE :129-135, metrics :140-142, time loop :144-181.
CDFVHST = r"""
SUBROUTINE lift_cdfvhst(kpi, kpj, kpk, kpt, kfull, kvvl, h_e1v, h_e2u, h_e31d, h_e3v, h_e3u, h_vt, h_vs, h_ut, h_us, o_var) &
     &  BIND(C, name='lift_cdfvhst')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kfull, kvvl
  REAL(C_FLOAT), INTENT(in) :: h_e1v(kpi,kpj,1,1), h_e2u(kpi,kpj,1,1), h_e31d(kpk), h_e3v(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_e3u(kpi,kpj,kpk,1+(kpt-1)*kvvl)
  REAL(C_FLOAT), INTENT(in) :: h_vt(kpi,kpj,kpk,kpt), h_vs(kpi,kpj,kpk,kpt), h_ut(kpi,kpj,kpk,kpt), h_us(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(out) :: o_var(kpi,kpj,4,kpt)          ! SNGL(dtrpvt), SNGL(dtrpvs), SNGL(dtrput), SNGL(dtrpus)
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 25 56
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; lfull = ( kfull /= 0 ) ; lg_vvl = ( kvvl /= 0 )
!@RAW 129 135
!@LIFT 140 142
!@LIFT 144 181
END SUBROUTINE lift_cdfvhst
"""


def gen_cdfvhst():
    t = Tool("cdfvhst.f90",
             {"cn_ve1v": "h_e1v", "cn_ve2u": "h_e2u", "cn_ve3t1d": "h_e31d", "cn_vomevt": "h_vt", "cn_vomevs": "h_vs",
              "cn_vozout": "h_ut", "cn_vozous": "h_us", "cn_ve3v": "h_e3v", "cn_ve3u": "h_e3u"},
             putvar=lambda pos, kw: f"o_var(:,:,{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFVHST)


# =============================================================================================== cdfvtrp
# src/cdfvtrp.f90: declarations :27-62, ALLOCATE :151-163, metrics :167-176, time loop :178-252 (with -bathy: the slope
# rotation through the compiler's ATAN2 / COS / SIN).
CDFVTRP = r"""
SUBROUTINE lift_cdfvtrp(kpi, kpj, kpk, kpt, kfull, kbathy, kvvl, h_e1v, h_e2u, h_e1u, h_e2v, h_tmask, h_hdepw, h_e31d, h_e3v, h_e3u, &
     &  h_u, h_v, o_var) BIND(C, name='lift_cdfvtrp')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kfull, kbathy, kvvl
  REAL(C_FLOAT), INTENT(in) :: h_e1v(kpi,kpj,1,1), h_e2u(kpi,kpj,1,1), h_e1u(kpi,kpj,1,1), h_e2v(kpi,kpj,1,1)
  REAL(C_FLOAT), INTENT(in) :: h_tmask(kpi,kpj,1,1), h_hdepw(kpi,kpj,1,1), h_e31d(kpk), h_e3v(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_e3u(kpi,kpj,kpk,1+(kpt-1)*kvvl)
  REAL(C_FLOAT), INTENT(in) :: h_u(kpi,kpj,kpk,kpt), h_v(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(out) :: o_var(kpi,kpj,4,kpt)          ! REAL(dtrpu), REAL(dtrpv) [, along-slope, cross-slope]
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 27 62
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; lfull = ( kfull /= 0 ) ; lbathy = ( kbathy /= 0 ) ; lg_vvl = ( kvvl /= 0 )
  o_var = 0.
!@RAW 151 163
!@LIFT 167 176
!@LIFT 178 252
END SUBROUTINE lift_cdfvtrp
"""


def gen_cdfvtrp():
    t = Tool("cdfvtrp.f90",
             {"cn_ve1v": "h_e1v", "cn_ve2u": "h_e2u", "cn_ve1u": "h_e1u", "cn_ve2v": "h_e2v", "cn_tmask": "h_tmask",
              "cn_hdepw": "h_hdepw", "cn_ve3t1d": "h_e31d", "cn_vozocrtx": "h_u", "cn_vomecrty": "h_v", "cn_ve3v": "h_e3v",
              "cn_ve3u": "h_e3u"},
             putvar=lambda pos, kw: f"o_var(:,:,{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFVTRP)


# =============================================================================================== cdfsum
# src/cdfsum.f90: declarations :29-79, ALLOCATE :214-217, metrics :275-286, time loop :290-349.  The PRINTs of the level
# sums (:324 / :327, forcing form :334) become the capture of dsurf, dvol2d, dsum2d.
CDFSUM = r"""
SUBROUTINE lift_cdfsum(kpi, kpj, kpk, kpt, kforcing, knwgh, knmsk, kvvl, h_e1, h_e2, h_dep, h_e3, h_msk, h_in, o_var, o_sums) &
     &  BIND(C, name='lift_cdfsum')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kforcing, knwgh, knmsk, kvvl
  REAL(C_FLOAT), INTENT(in) :: h_e1(kpi,kpj,1,1), h_e2(kpi,kpj,1,1), h_dep(kpk), h_e3(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_msk(kpi,kpj,kpk,1)
  REAL(C_FLOAT), INTENT(in) :: h_in(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT),  INTENT(out) :: o_var(kpk,3,kpt)            ! rdummy as handed to putvar: level sums, cumulated, per record
  REAL(C_DOUBLE), INTENT(out) :: o_sums(3,kpk,kpt)           ! dsurf, dvol2d, dsum2d of each level
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 29 79
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; npkk = kpk ; nvpk = kpk
  iimin = 1 ; ijmin = 1 ; ikmin = 1 ; ikmax = kpk             ! no -w window: src/cdfsum.f90:190-192
  lforcing = ( kforcing /= 0 ) ; lnwgh = ( knwgh /= 0 ) ; lnmsk = ( knmsk /= 0 ) ; lg_vvl = ( kvvl /= 0 )
  o_var = 0. ; o_sums = 0.d0
!@RAW 214 217
!@LIFT 275 286
!@LIFT 290 349
END SUBROUTINE lift_cdfsum
"""


def gen_cdfsum():
    cap = "           o_sums(1,ik,jt) = dsurf ; o_sums(2,ik,jt) = dvol2d ; o_sums(3,ik,jt) = dsum2d   ! [lift] capture of the PRINT"
    capf = "           o_sums(1,ik,jt) = dsurf ; o_sums(3,ik,jt) = dsum2d   ! [lift] capture of the PRINT"
    t = Tool("cdfsum.f90",
             {"cv_e1": "h_e1", "cv_e2": "h_e2", "cv_dep": "h_dep", "cv_in": "h_in", "cv_msk": "h_msk", "cv_e3": "h_e3"},
             putvar=lambda pos, kw: f"o_var({pos[3]}:{pos[3]},{varidx(pos[1])}:{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}",
             capture={324: cap, 327: cap, 334: capf})
    return t, t.expand(CDFSUM)


# =============================================================================================== cdfmean
# src/cdfmean.f90: declarations :27-109, ALLOCATE :294-303, metrics :361-366, :376-378, basin masks :383-389, time loop
# :396-482.  The WRITEs of the level values (:429 / :440) become the capture of dsurf, dvol2d, dsum2d, dvar2d.
CDFMEAN = r"""
SUBROUTINE lift_cdfmean(kpi, kpj, kpk, kpt, kbasin, kfull, kvar, ksum, kvvl, h_e1, h_e2, h_e31d, h_dep, h_e3, h_msk, h_bas, h_in, &
     &  knv, o_var, o_var0, o_sums) BIND(C, name='lift_cdfmean')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kbasin, kfull, kvar, ksum, knv, kvvl
  REAL(C_FLOAT), INTENT(in) :: h_e1(kpi,kpj,1,1), h_e2(kpi,kpj,1,1), h_e31d(kpk), h_dep(kpk), h_e3(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_msk(kpi,kpj,kpk,1)
  REAL(C_FLOAT), INTENT(in) :: h_bas(kpi,kpj,kbasin,1), h_in(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT),  INTENT(out) :: o_var(kpk,knv,kpt)          ! rdummymean as handed to putvar, per output variable index
  REAL(C_FLOAT),  INTENT(out) :: o_var0(knv,kpt)             ! rdummy as handed to putvar0d
  REAL(C_DOUBLE), INTENT(out) :: o_sums(4,kbasin,kpk,kpt)    ! dsurf, dvol2d, dsum2d, dvar2d of each level and basin
  LOGICAL :: lg_vvl = .FALSE.
  INTEGER :: jb_
!@DECLS 27 109
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; nvpk = kpk ; npk_fi = kpk ; nbasin = kbasin
  iimin = 1 ; ijmin = 1 ; ikmin = 1 ; ikmax = kpk             ! no -w window: src/cdfmean.f90:268-270
  lbas = ( kbasin > 1 ) ; lfull = ( kfull /= 0 ) ; lvar = ( kvar /= 0 ) ; lsum = ( ksum /= 0 ) ; lg_vvl = ( kvvl /= 0 )
  o_var = 0. ; o_var0 = 0. ; o_sums = 0.d0
!@RAW 294 303
!@LIFT 361 366
!@RAW 368 373
  ALLOCATE ( ipk(nvars), id_varout(nvars) )
!@RAW 376 378
  ! output variable numbers as CreateOutput hands them out (src/cdfmean.f90:556-620): per basin mean, mean3d [, var, var3d] [, sum, sum3d]
  DO jb_ = 1, nbasin
     ivarb = nvars_bas * (jb_ - 1)
     n_mean(jb_) = ivarb + 1 ; n_mean3d(jb_) = ivarb + 2
     ivar = 2
     IF ( lvar ) THEN ; n_var(jb_) = ivarb + ivar + 1 ; n_var3d(jb_) = ivarb + ivar + 2 ; ivar = ivar + 2 ; ENDIF
     IF ( lsum ) THEN ; n_sum(jb_) = ivarb + ivar + 1 ; n_sum3d(jb_) = ivarb + ivar + 2 ; ENDIF
  ENDDO
  ALLOCATE ( cbasins(nbasin) ) ; cbasins = 'basin'
!@LIFT 383 389
!@LIFT 396 482
END SUBROUTINE lift_cdfmean
"""


def gen_cdfmean():
    cap = ("              o_sums(1,jbasin,jk,jt) = dsurf ; o_sums(2,jbasin,jk,jt) = dvol2d ; o_sums(3,jbasin,jk,jt) = dsum2d ; "
           "o_sums(4,jbasin,jk,jt) = dvar2d   ! [lift] capture of the WRITE")
    t = Tool("cdfmean.f90",
             {"cv_e1": "h_e1", "cv_e2": "h_e2", "cv_e31d": "h_e31d", "cv_dep": "h_dep", "cv_nam": "h_in", "cv_msk": "h_msk",
              "cv_e3": "h_e3", "cbasins(jbasin)": "h_bas(:,:,jbasin,1)"},
             putvar=lambda pos, kw: f"o_var({pos[3]}:{pos[3]},{varidx(pos[1])}:{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}",
             capture={429: cap, 440: cap})
    t.putvar0d = lambda pos, kw: f"o_var0({varidx(pos[1])}:{varidx(pos[1])},{kw.get('ktime', '1')}:{kw.get('ktime', '1')}) = {pos[2]}"
    return t, t.expand(CDFMEAN)


# =============================================================================================== cdfvertmean
# src/cdfvertmean.f90: declarations :28-79, ALLOCATE :183-188, depths :210-218, time loop :224-269.
CDFVERTMEAN = r"""
SUBROUTINE lift_cdfvertmean(kpi, kpj, kpk, kpt, kfull, kvvl, pup, pdown, h_dep, h_e31d, h_e3, h_msk, h_in, o_var) &
     &  BIND(C, name='lift_cdfvertmean')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kfull, kvvl
  REAL(C_FLOAT), VALUE :: pup, pdown
  REAL(C_FLOAT), INTENT(in) :: h_dep(kpk), h_e31d(kpk), h_e3(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_msk(kpi,kpj,kpk,1), h_in(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(out) :: o_var(kpi,kpj,1,kpt)          ! REAL(dvertmean) as handed to putvar
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 28 79
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; nvaro = 1 ; iimin = 1 ; ijmin = 1
  lfull = ( kfull /= 0 ) ; lg_vvl = ( kvvl /= 0 ) ; rdep_up = pup ; rdep_down = pdown
  ALLOCATE ( cv_in(1), ipk(1), id_varout(1) ) ; cv_in(1) = 'field'
!@RAW 183 188
!@LIFT 210 218
!@LIFT 224 269
END SUBROUTINE lift_cdfvertmean
"""


def gen_cdfvertmean():
    t = Tool("cdfvertmean.f90",
             {"cv_dep": "h_dep", "cv_e31d": "h_e31d", "cv_cur": "h_in", "cv_msk": "h_msk", "cv_e3": "h_e3"},
             putvar=lambda pos, kw: f"o_var(:,:,{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFVERTMEAN)


# =============================================================================================== cdfvint
# src/cdfvint.f90: declarations :28-68, ALLOCATE :234-242, depths :245-259, time loop :268-313.
CDFVINT = r"""
SUBROUTINE lift_cdfvint(kpi, kpj, kpk, kpt, kfull, kgsop, kocci, kmean, kvvl, psclf, h_gdepw, h_e31d, h_tmask, h_e3t, h_in, knpko, o_var) &
     &  BIND(C, name='lift_cdfvint')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kfull, kgsop, kocci, kmean, knpko, kvvl
  REAL(C_FLOAT), VALUE :: psclf
  REAL(C_FLOAT), INTENT(in) :: h_gdepw(kpk), h_e31d(kpk), h_tmask(kpi,kpj,kpk,1), h_e3t(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_in(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(out) :: o_var(kpi,kpj,knpko+1,kpt)    ! REAL(dl_vint2) [/rdep2] as handed to putvar, per output depth
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 28 68
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; npko = knpko ; iimin = 1 ; ijmin = 1
  lfull = ( kfull /= 0 ) ; lgsop = ( kgsop /= 0 ) ; locci = ( kocci /= 0 ) ; ltmean = ( kmean /= 0 ) ; lsmean = .FALSE.
  sclf = psclf ; lg_vvl = ( kvvl /= 0 )
  o_var = 0.
!@RAW 234 242
!@LIFT 245 259
!@LIFT 268 313
END SUBROUTINE lift_cdfvint
"""


def gen_cdfvint():
    t = Tool("cdfvint.f90",
             {"cn_gdepw": "h_gdepw", "cn_ve3t1d": "h_e31d", "cn_tmask": "h_tmask", "cv_in": "h_in", "cn_ve3t": "h_e3t"},
             putvar=lambda pos, kw: f"o_var(:,:,{pos[3]},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFVINT)


# =============================================================================================== cdfzonalsum
# src/cdfzonalsum.f90: declarations :27-83, ALLOCATE :246-252, metrics / alpha / depths :255-280, basin masks :286-294,
# main loop :297-326 (one variable of npk levels).
CDFZONALSUM = r"""
SUBROUTINE lift_cdfzonalsum(kpi, kpj, kpk, kpt, kbas, kpdeg, h_e1, h_e2, h_phi, h_dep, h_msk, h_atl, h_ind, h_pac, h_in, o_var, o_r8) &
     &  BIND(C, name='lift_cdfzonalsum')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kbas, kpdeg
  REAL(C_FLOAT), INTENT(in) :: h_e1(kpi,kpj,1,1), h_e2(kpi,kpj,1,1), h_phi(kpi,kpj,1,1), h_dep(kpk), h_msk(kpi,kpj,kpk,1)
  REAL(C_FLOAT), INTENT(in) :: h_atl(kpi,kpj,1,1), h_ind(kpi,kpj,1,1), h_pac(kpi,kpj,1,1), h_in(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT),  INTENT(out) :: o_var(kpj,kpk,kbas,kpt)     ! REAL(dzosum(:,jk)) as handed to putvar
  REAL(C_DOUBLE), INTENT(out) :: o_r8(kpj,kpk,kbas,kpt)      ! dzosum(:,jk) itself
!@DECLS 27 83
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; npbasins = kbas ; lpdeg = ( kpdeg /= 0 ) ; l2d = .FALSE.
  nvar = 1 ; nvaro = 1 ; nvarin = 1 ; nvarmx = npbasins
  cf_basins = 'none' ; IF ( kbas == 5 ) cf_basins = 'yes'        ! (initialised variables are SAVEd: set both ways)
  ALLOCATE ( cv_namesi(1), ipki(1), id_varin(1), ipko(nvarmx), id_varout(nvarmx) )
  cv_namesi(1) = 'field' ; ipki(1) = npk ; id_varin(1) = 1
!@RAW 246 252
!@LIFT 255 280
!@LIFT 286 294
!@LIFT 297 326
END SUBROUTINE lift_cdfzonalsum
"""


def gen_cdfzonalsum():
    t = Tool("cdfzonalsum.f90",
             {"cv_e1": "h_e1", "cv_e2": "h_e2", "cv_phi": "h_phi", "cv_depi": "h_dep", "cv_msk": "h_msk", "cn_tmaskatl": "h_atl",
              "cn_tmaskind": "h_ind", "cn_tmaskpac": "h_pac", "cv_namesi(ijvar)": "h_in"},
             putvar=lambda pos, kw: (f"o_var(:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]} ; "
                                     f"o_r8(:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = dzosum(:,jk)"))
    return t, t.expand(CDFZONALSUM)


# =============================================================================================== cdfzonalmean
# src/cdfzonalmean.f90: declarations :27-84, ALLOCATE :248-256, metrics / depths :259-278, basin masks :287-297, main loop
# :300-350 (one variable of npk levels; -max: extrema).
CDFZONALMEAN = r"""
SUBROUTINE lift_cdfzonalmean(kpi, kpj, kpk, kpt, kbas, kmax, h_e1, h_e2, h_phi, h_dep, h_msk, h_atl, h_ind, h_pac, h_in, o_var, o_r8, &
     &  o_area) BIND(C, name='lift_cdfzonalmean')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kbas, kmax
  REAL(C_FLOAT), INTENT(in) :: h_e1(kpi,kpj,1,1), h_e2(kpi,kpj,1,1), h_phi(kpi,kpj,1,1), h_dep(kpk), h_msk(kpi,kpj,kpk,1)
  REAL(C_FLOAT), INTENT(in) :: h_atl(kpi,kpj,1,1), h_ind(kpi,kpj,1,1), h_pac(kpi,kpj,1,1), h_in(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT),  INTENT(out) :: o_var(kpj,kpk,3*kbas,kpt)   ! REAL(dzomean(:,jk)) [, rzomax, rzomin] as handed to putvar
  REAL(C_DOUBLE), INTENT(out) :: o_r8(kpj,kpk,kbas,kpt)      ! dzomean(:,jk) after the division
  REAL(C_DOUBLE), INTENT(out) :: o_area(kpj,kpk,kbas,kpt)    ! darea(:,jk)
!@DECLS 27 84
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; npbasins = kbas ; lmax = ( kmax /= 0 ) ; l2d = .FALSE.
  nvar = 1 ; nvarin = 1 ; nvaro = npbasins ; nvarmx = 3 * npbasins          ! nvaro as CreateOutput leaves it (the count of mean variables): src/cdfzonalmean.f90:448
  cf_basins = 'none' ; IF ( kbas == 5 ) cf_basins = 'yes'        ! (initialised variables are SAVEd: set both ways)
  ALLOCATE ( cv_namesi(1), ipki(1), id_varin(1), ipko(nvarmx), id_varout(nvarmx) )
  cv_namesi(1) = 'field' ; ipki(1) = npk ; id_varin(1) = 1
  o_var = 0.
!@RAW 248 256
!@LIFT 259 278
!@LIFT 287 297
!@LIFT 300 350
END SUBROUTINE lift_cdfzonalmean
"""


def gen_cdfzonalmean():
    def put(pos, kw):
        base = f"o_var(:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}"
        if "dzomean" in pos[2]:
            base += (f" ; o_r8(:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = dzomean(:,jk)"
                     f" ; o_area(:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = darea(:,jk)")
        return base
    t = Tool("cdfzonalmean.f90",
             {"cv_e1": "h_e1", "cv_e2": "h_e2", "cv_phi": "h_phi", "cv_depi": "h_dep", "cv_msk": "h_msk", "cn_tmaskatl": "h_atl",
              "cn_tmaskind": "h_ind", "cn_tmaskpac": "h_pac", "cv_namesi(ijvar)": "h_in"},
             putvar=put)
    return t, t.expand(CDFZONALMEAN)


# =============================================================================================== cdfmocsig
# src/cdfmocsig.f90: declarations :40-108, sizes :255-261, bins :299-304, ALLOCATE :308-323, metrics :325-328, basin numbers
# :342, masks :347-359, the whole time loop :361-485 (density through the reference's eos module; -isodep, -eiv, -full, -ntr).
CDFMOCSIG = r"""
SUBROUTINE lift_cdfmocsig(kpi, kpj, kpk, kpt, kbas, kfull, kisodep, keiv, kntr, kteos10, knbins, kvvl, psigmin, psigstp, ppref, pspv, &
     &  pspt, psps, h_e1v, h_e31d, h_gdept, h_vmask, h_atl, h_ind, h_pac, h_e3v, h_v, h_veiv, h_t, h_s, o_var, o_dmoc, o_depi) &
     &  BIND(C, name='lift_cdfmocsig')
  USE ISO_C_BINDING
  USE eos                      ! the reference's src/eos.f90, compiled where it lies
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kbas, kfull, kisodep, keiv, kntr, kteos10, knbins, kvvl
  REAL(C_FLOAT), VALUE :: psigmin, psigstp, ppref, pspv, pspt, psps
  REAL(C_FLOAT), INTENT(in) :: h_e1v(kpi,kpj,1,1), h_e31d(kpk), h_gdept(kpk), h_vmask(kpi,kpj,1,1), h_atl(kpi,kpj,1,1)
  REAL(C_FLOAT), INTENT(in) :: h_ind(kpi,kpj,1,1), h_pac(kpi,kpj,1,1), h_e3v(kpi,kpj,kpk,1+(kpt-1)*kvvl), h_v(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(in) :: h_veiv(kpi,kpj,kpk,kpt), h_t(kpi,kpj,kpk,kpt), h_s(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT),  INTENT(out) :: o_var(kpj,knbins,2*kbas,kpt)     ! REAL(dmoc(basin,bin,:)) [, REAL(depi(..))] as handed to putvar
  REAL(C_DOUBLE), INTENT(out) :: o_dmoc(kbas,knbins,kpj,kpt), o_depi(kbas,knbins,kpj,kpt)
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 40 108
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt
  lbas = ( kbas == 5 ) ; lfull = ( kfull /= 0 ) ; lisodep = ( kisodep /= 0 ) ; leiv = ( keiv /= 0 ) ; lntr = ( kntr /= 0 )
  ll_teos10 = ( kteos10 /= 0 ) ; lprint = .FALSE. ; lg_vvl = ( kvvl /= 0 )
  nbins = knbins ; sigmin = psigmin ; sigstp = psigstp ; pref = ppref ; zspv = pspv ; zspt = pspt ; zsps = psps
  CALL eos_init ( ll_teos10 )                              ! src/cdfmocsig.f90:210
  o_var = 0. ; o_dmoc = 0.d0 ; o_depi = 0.d0
!@RAW 255 261
  ALLOCATE ( ipk(nvaro), id_varout(nvaro) )
!@RAW 299 304
!@RAW 308 323
!@LIFT 325 328
!@RAW 342 342
!@LIFT 347 359
!@LIFT 361 476
     o_dmoc(:,:,:,jt) = dmoc(1:nbasins,:,:)
     IF ( lisodep ) o_depi(:,:,:,jt) = depi(1:nbasins,:,:)
!@LIFT 477 485
END SUBROUTINE lift_cdfmocsig
"""


def gen_cdfmocsig():
    t = Tool("cdfmocsig.f90",
             {"cn_ve1v": "h_e1v", "cn_ve3t1d": "h_e31d", "cn_gdept": "h_gdept", "cn_gphiv": "h_e1v", "cn_vmask": "h_vmask",
              "cn_tmaskatl": "h_atl", "cn_tmaskind": "h_ind", "cn_tmaskpac": "h_pac", "cn_vomecrty": "h_v", "cn_vomeeivv": "h_veiv",
              "cn_votemper": "h_t", "cn_vosaline": "h_s", "cn_ve3v": "h_e3v"},
             putvar=lambda pos, kw: f"o_var(:,{pos[3]},{varidx(pos[1])},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFMOCSIG)


# =============================================================================================== cdfmoc -rapid
# src/cdfmoc.f90, SUBROUTINE rapid_amoc (:598-1004), contained in the program: its own declarations :618-653 (the USE of
# cdftools, :617, is for cdf_findij, which the harness replaces by the section's indices), ALLOCATE :683-690, the metric /
# mask reads :692-705, the whole time loop :709-998 (33 output variables).  Kept as an internal subroutine of the harness so
# that its local gdept hides the program's, as in the reference.
CDFMOC_RAPID = r"""
SUBROUTINE lift_cdfmoc_rapid(kpi, kpigs, kpk, kpt, kfull, kvvl, h_gphiv, h_e31d, h_e1v, h_gdepw, h_tmask, h_gdept, h_e3v, h_v, h_taux, &
     &  h_t, h_s, o_var) BIND(C, name='lift_cdfmoc_rapid')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpigs, kpk, kpt, kfull, kvvl    ! kpi: cells of the section (iiw..iie), kpigs: of its Gulf Stream part
  REAL(C_FLOAT), INTENT(in) :: h_gphiv(kpi,1,1,1), h_e31d(kpk), h_e1v(kpi,1,1,1), h_gdepw(kpk), h_tmask(kpi,1,kpk,1), h_gdept(kpk)
  REAL(C_FLOAT), INTENT(in) :: h_e3v(kpi,1,kpk,1+(kpt-1)*kvvl), h_v(kpi,1,kpk,kpt), h_taux(kpi,1,1,kpt), h_t(kpi,1,kpk,kpt), h_s(kpi,1,kpk,kpt)
  REAL(C_FLOAT), INTENT(out) :: o_var(33,kpk,kpt)                ! what putvar is handed, by output variable and level
  LOGICAL :: lg_vvl = .FALSE.
!@DECLS 49 125
  npk = kpk ; npt = kpt ; lfull = ( kfull /= 0 ) ; lg_vvl = ( kvvl /= 0 )
  o_var = 0.
  CALL rapid_amoc
CONTAINS
  SUBROUTINE rapid_amoc
!@DECLS 618 653
    npiglo = kpi ; npigs = kpigs
!@RAW 683 690
!@LIFT 692 705
!@LIFT 709 998
  END SUBROUTINE rapid_amoc
END SUBROUTINE lift_cdfmoc_rapid
"""


def gen_cdfmoc_rapid():
    def put(pos, kw):
        idx, lev = varidx(pos[1]), pos[3]
        return f"o_var({idx}:{idx},{lev}:{lev},{kw.get('ktime', '1')}) = {pos[2]}"
    t = Tool("cdfmoc.f90",
             {"cn_gphiv": "h_gphiv", "cn_ve3t1d": "h_e31d", "cn_ve1v": "h_e1v", "cn_gdepw": "h_gdepw", "cn_tmask": "h_tmask",
              "cn_gdept": "h_gdept", "cn_ve3v": "h_e3v", "cn_vomecrty": "h_v", "cn_sozotaux": "h_taux", "cn_votemper": "h_t",
              "cn_vosaline": "h_s"}, putvar=put)
    return t, t.expand(CDFMOC_RAPID)


# =============================================================================================== cdfmaxmoc
# src/cdfmaxmoc.f90 (the consumer of moc.nc): declarations :28-59, ALLOCATE :122, depths and latitudes :123-124, the read of
# the overturning and the window search :140-178 (six scalars to putvar0d).
CDFMAXMOC = r"""
SUBROUTINE lift_cdfmaxmoc(kpj, kpk, platmin, platmax, pdepmin, pdepmax, h_depw, h_lat, h_moc, o_var) BIND(C, name='lift_cdfmaxmoc')
  USE ISO_C_BINDING
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpj, kpk
  REAL(C_FLOAT), VALUE :: platmin, platmax, pdepmin, pdepmax
  REAL(C_DOUBLE), INTENT(in) :: h_depw(kpk)                     ! depthw of moc.nc as getvar1d returns it (r8, negative downwards)
  REAL(C_FLOAT), INTENT(in) :: h_lat(1,kpj,1,1), h_moc(1,kpj,kpk,1)
  REAL(C_FLOAT), INTENT(out) :: o_var(6)                        ! maxmoc, minmoc, their latitudes, their depths
!@DECLS 28 59
  npjglo = kpj ; npk = kpk ; rlatmin = platmin ; rlatmax = platmax ; rdepmin = pdepmin ; rdepmax = pdepmax
!@RAW 122 122
!@LIFT 123 124
!@LIFT 140 178
END SUBROUTINE lift_cdfmaxmoc
"""


def gen_cdfmaxmoc():
    t = Tool("cdfmaxmoc.f90", {"cn_vdepthw": "h_depw", "cn_vlat2d": "h_lat", "cv_in": "h_moc"})
    t.putvar0d = lambda pos, kw: f"o_var({varidx(pos[1])}) = {pos[2]}"
    return t, t.expand(CDFMAXMOC)


# =============================================================================================== cdfsig0, cdfsigi
# The two thin callers of the EOS: src/cdfsig0.f90 declarations :25-51, ALLOCATE :132-134, the
# time loop :139-155; src/cdfsigi.f90 declarations :26-52, ALLOCATE :124-126, the time loop :130-145.  Density through the
# reference's eos module, as compiled where it lies.
CDFSIG0 = r"""
SUBROUTINE lift_cdfsig0(kpi, kpj, kpk, kpt, kteos10, psps, h_t, h_s, o_var) BIND(C, name='lift_cdfsig0')
  USE ISO_C_BINDING
  USE eos                      ! the reference's src/eos.f90, compiled where it lies
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kteos10
  REAL(C_FLOAT), VALUE :: psps
  REAL(C_FLOAT), INTENT(in) :: h_t(kpi,kpj,kpk,kpt), h_s(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(out) :: o_var(kpi,kpj,kpk,kpt)        ! zsig0 as handed to putvar
!@DECLS 25 51
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npkk = kpk ; npt = kpt ; ll_teos10 = ( kteos10 /= 0 ) ; zsps = psps
  CALL eos_init ( ll_teos10 )                              ! src/cdfsig0.f90:110
!@RAW 132 134
  dtim(:) = 0.d0
!@LIFT 139 155
END SUBROUTINE lift_cdfsig0
"""


def gen_cdfsig0():
    t = Tool("cdfsig0.f90", {"cv_tem": "h_t", "cv_sal": "h_s"},
             putvar=lambda pos, kw: f"o_var(:,:,{pos[3]},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFSIG0)


CDFSIGI = r"""
SUBROUTINE lift_cdfsigi(kpi, kpj, kpk, kpt, kteos10, pref, pspval, h_t, h_s, o_var) BIND(C, name='lift_cdfsigi')
  USE ISO_C_BINDING
  USE eos                      ! the reference's src/eos.f90, compiled where it lies
  IMPLICIT NONE
  INTEGER(C_INT), VALUE :: kpi, kpj, kpk, kpt, kteos10
  REAL(C_FLOAT), VALUE :: pref, pspval
  REAL(C_FLOAT), INTENT(in) :: h_t(kpi,kpj,kpk,kpt), h_s(kpi,kpj,kpk,kpt)
  REAL(C_FLOAT), INTENT(out) :: o_var(kpi,kpj,kpk,kpt)        ! zsigi as handed to putvar
!@DECLS 26 52
  npiglo = kpi ; npjglo = kpj ; npk = kpk ; npt = kpt ; ll_teos10 = ( kteos10 /= 0 ) ; ref_dep = pref ; zspval = pspval
  CALL eos_init ( ll_teos10 )                              ! src/cdfsigi.f90:108
!@RAW 124 126
!@LIFT 130 145
END SUBROUTINE lift_cdfsigi
"""


def gen_cdfsigi():
    t = Tool("cdfsigi.f90", {"cn_votemper": "h_t", "cn_vosaline": "h_s"},
             putvar=lambda pos, kw: f"o_var(:,:,{pos[3]},{kw.get('ktime', '1')}) = {pos[2]}")
    return t, t.expand(CDFSIGI)


TOOLS = {"cdfmoc": gen_cdfmoc, "cdfheatc": gen_cdfheatc, "cdfmhst": gen_cdfmhst, "cdfpsi": gen_cdfpsi, "cdfvt": gen_cdfvt,
         "cdfcurl": gen_cdfcurl, "cdfvhst": gen_cdfvhst, "cdfvtrp": gen_cdfvtrp, "cdfsum": gen_cdfsum,
         "cdfmean": gen_cdfmean, "cdfvertmean": gen_cdfvertmean, "cdfvint": gen_cdfvint,
         "cdfzonalsum": gen_cdfzonalsum, "cdfzonalmean": gen_cdfzonalmean, "cdfmocsig": gen_cdfmocsig,
         "cdfsig0": gen_cdfsig0, "cdfsigi": gen_cdfsigi, "cdfmoc_rapid": gen_cdfmoc_rapid,
         "cdfmaxmoc": gen_cdfmaxmoc}


def main():
    global OUT
    if "--out" in sys.argv:
        OUT = sys.argv[sys.argv.index("--out") + 1]
    os.makedirs(OUT, exist_ok=True)
    for name, gen in TOOLS.items():
        tool, text = gen()
        path = os.path.join(OUT, f"lift_{name}.f90")
        with open(path, "w") as f:
            f.write(f"! GENERATED by oracle/ref_lift.py from {REF}/{tool.fname} -- never committed (oracle/_ref is git-ignored)\n")
            f.write(text)
        if "--shas" in sys.argv:
            print(name, {k: v for k, v in tool.report})
    return 0


if __name__ == "__main__":
    sys.exit(main())
