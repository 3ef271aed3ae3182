"""The program-level differential (tests/make_golden_prog.py writes, tests/test_prog_diff.py reads): the cases -- the eight
programs of SURVEY 8(a) and the ten of 8(f), each with its default call and with one option -- and the dump of everything a run leaves behind
(exit status, stdout, every output file's dimensions, variables, attributes in file order, data, text dumps).

A case runs in a directory holding the files of nemo_files.make_case (24 x 18 x 7, two records, five basins); cases run in
the order listed, in ONE directory, because cdfmhst reads the vt.nc that cdfvT wrote."""
import json
import os
import subprocess

import numpy as np
from scipy.io import netcdf_file

import run_program

UV = ["-u", "SYN_y0001_gridU.nc", "-v", "SYN_y0001_gridV.nc"]

# (case name, program, arguments, output files)
CASES = [
    ("cdfmoc_default", "cdfmoc", ["-v", "SYN_y0001_gridV.nc"], ["moc.nc"]),
    ("cdfmoc_full", "cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-full", "-o", "moc_full.nc"], ["moc_full.nc"]),
    ("cdfheatc_default", "cdfheatc", ["-f", "SYN_y0001_gridT.nc"], ["heatc.nc"]),
    ("cdfheatc_zoom_full", "cdfheatc", ["-f", "SYN_y0001_gridT.nc", "-zoom", "3", "12", "2", "9", "2", "4", "-full", "-o", "heatc_z.nc"], ["heatc_z.nc"]),
    ("cdfcurl_default", "cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "1"], ["curl.nc"]),
    ("cdfcurl_levels_T", "cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "2-4", "-T",
                          "-o", "curl_t.nc"], ["curl_t.nc"]),
    ("cdfcurl_overf_8", "cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "1,3", "-overf", "-8",
                         "-o", "curl_f.nc"], ["curl_f.nc"]),
    ("cdfsig0_default", "cdfsig0", ["-t", "SYN_y0001_gridT.nc"], ["sig0.nc"]),
    ("cdfsig0_teos10", "cdfsig0", ["-t", "SYN_y0001_gridT.nc", "-teos10", "-o", "sig0_teos.nc"], ["sig0_teos.nc"]),
    ("cdfsigi_default", "cdfsigi", ["-t", "SYN_y0001_gridT.nc", "-r", "2000"], ["sigi.nc"]),
    ("cdfsigi_teos10", "cdfsigi", ["-t", "SYN_y0001_gridT.nc", "-r", "1000", "-teos10", "-o", "sigi_teos.nc"], ["sigi_teos.nc"]),
    ("cdfpsi_default", "cdfpsi", UV, ["psi.nc"]),
    ("cdfpsi_V_mask_ref", "cdfpsi", UV + ["-V", "-mask", "-ref", "5", "4", "-o", "psi_v.nc"], ["psi_v.nc"]),
    ("cdfvT_default", "cdfvT", ["-c", "SYN", "-l", "y0001"], ["vt.nc"]),
    ("cdfvT_named_output", "cdfvT", ["-c", "SYN", "-l", "y0001", "-namT", "votemper", "-o", "vt_named.nc"], ["vt_named.nc"]),
    ("cdfmhst_default", "cdfmhst", ["-vt", "vt.nc"], ["mhst.nc", "zonal_heat_trp.dat", "zonal_salt_trp.dat"]),
    ("cdfmhst_MST_Zdim", "cdfmhst", ["-vt", "vt.nc", "-MST", "-Zdim", "-o", "mhst_z.nc"], ["mhst_z.nc", "zonal_heat_trp.dat", "zonal_salt_trp.dat"]),
    # more options of the eight (round 4, second tier); extra_files() writes mxl.nc and SYN_y0001_grid2D.nc
    ("cdfmoc_decomp", "cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-decomp", "-o", "moc_dec.nc"], ["moc_dec.nc"]),
    ("cdfmoc_teos10_decomp", "cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-decomp", "-teos10", "-o", "moc_dect.nc"], ["moc_dect.nc"]),
    ("cdfmoc_teos_as_the_usage_spells_it", "cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-decomp", "-teos"], []),   # (the parser knows -teos10: STOP 99)
    ("cdfheatc_mxl_in", "cdfheatc", ["-f", "SYN_y0001_gridT.nc", "-mxloption", "1", "-mxlf", "mxl.nc", "-o", "heatc_m.nc"], ["heatc_m.nc"]),
    ("cdfheatc_mxl_out_M", "cdfheatc", ["-f", "SYN_y0001_gridT.nc", "-mxloption", "-1", "-mxlf", "mxl.nc", "-M", "mask.nc", "umask", "-o", "heatc_mo.nc"],
     ["heatc_mo.nc"]),
    ("cdfcurl_surf", "cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "1", "-surf", "-o", "curl_s.nc"],
     ["curl_s.nc"]),
    ("cdfpsi_mean", "cdfpsi", UV + ["-mean", "-o", "psi_m.nc"], ["psi_m.nc"]),
    ("cdfpsi_open_ref_mask", "cdfpsi", UV + ["-open", "-ref", "5", "4", "-mask", "-o", "psi_o.nc"], ["psi_o.nc"]),
    ("cdfpsi_ssh", "cdfpsi", UV + ["--ssh-file", "SYN_y0001_grid2D.nc", "-o", "psi_s.nc"], ["psi_s.nc"]),
    ("cdfpsi_ssh_mean", "cdfpsi", UV + ["-mean", "--ssh-file", "SYN_y0001_grid2D.nc", "-o", "psi_sm.nc"], ["psi_sm.nc"]),
    ("cdfpsi_lev", "cdfpsi", UV + ["-lev", "-o", "psi_l.nc"], ["psi_l.nc"]),
    ("cdfpsi_full", "cdfpsi", UV + ["-full", "-o", "psi_f.nc"], ["psi_f.nc"]),
    ("cdfmhst_VTS_files", "cdfmhst", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-MST", "-o", "mhst_s.nc"],
     ["mhst_s.nc", "zonal_heat_trp.dat", "zonal_salt_trp.dat"]),
    ("cdfmhst_full", "cdfmhst", ["-vt", "vt.nc", "-full", "-o", "mhst_f.nc"], ["mhst_f.nc"]),
    ("cdfsig0_named", "cdfsig0", ["-t", "SYN_y0001_gridT.nc", "-s", "SYN_y0001_gridT.nc", "-sal", "vosaline", "-tem", "votemper", "-o", "sig0_n.nc"],
     ["sig0_n.nc"]),
    ("cdfsigi_sfile_r_last", "cdfsigi", ["-t", "SYN_y0001_gridT.nc", "-s", "SYN_y0001_gridT.nc", "-o", "sigi_s.nc", "-r", "500"], ["sigi_s.nc"]),
    # third tier: -vvl (per-record vertical metrics read from the data files) and more options of the 8f programs
    ("cdfmoc_vvl", "cdfmoc", ["-v", "VVL_y0001_gridV.nc", "-vvl", "-o", "moc_vvl.nc"], ["moc_vvl.nc"]),
    ("cdfheatc_vvl", "cdfheatc", ["-f", "VVL_y0001_gridT.nc", "-vvl", "-o", "heatc_vvl.nc"], ["heatc_vvl.nc"]),
    ("cdfmhst_vvl", "cdfmhst", ["-vt", "VVL_vt.nc", "-MST", "-vvl", "-o", "mhst_vvl.nc"], ["mhst_vvl.nc"]),
    ("cdfpsi_vvl", "cdfpsi", ["-u", "VVL_y0001_gridU.nc", "-v", "VVL_y0001_gridV.nc", "-vvl", "-o", "psi_vvl.nc"], ["psi_vvl.nc"]),
    ("cdfvhst_vvl", "cdfvhst", ["-f", "VVL_vt.nc", "-vvl", "-o", "vh_vvl.nc"], ["vh_vvl.nc"]),
    ("cdfvtrp_vvl", "cdfvtrp", ["-u", "VVL_y0001_gridU.nc", "-v", "VVL_y0001_gridV.nc", "-vvl", "-o", "tr_vvl.nc"], ["tr_vvl.nc"]),
    ("cdfvtrp_bathy", "cdfvtrp", UV + ["-bathy", "-o", "tr_b.nc"], ["tr_b.nc"]),
    ("cdfsum_vvl", "cdfsum", ["-f", "VVL_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-vvl", "-o", "sum_vvl.nc"], ["sum_vvl.nc"]),
    ("cdfsum_nomsk", "cdfsum", ["-f", "SYN_y0001_gridT.nc", "-v", "vosaline", "-p", "T", "-nomsk", "-o", "n.nc"], ["n.nc"]),
    ("cdfvint_vvl", "cdfvint", ["-f", "VVL_y0001_gridT.nc", "-vvl", "-o", "vint_vvl.nc"], ["vint_vvl.nc"]),
    ("cdfvint_GSOP_tmean", "cdfvint", ["-f", "SYN_y0001_gridT.nc", "-GSOP", "-tmean", "-o", "vint_g.nc"], ["vint_g.nc"]),
    ("cdfvertmean_vvl", "cdfvertmean", ["-f", "VVL_y0001_gridT.nc", "-l", "votemper", "-p", "T", "-zlim", "15", "120", "-vvl", "-o", "vm_vvl.nc"],
     ["vm_vvl.nc"]),
    ("cdfvertmean_full", "cdfvertmean", ["-f", "SYN_y0001_gridT.nc", "-l", "vosaline", "-p", "T", "-zlim", "0", "300", "-full", "-o", "vm_f.nc"], ["vm_f.nc"]),
    ("cdfmocsig_ntr_vvl", "cdfmocsig", ["-v", "VVL_y0001_gridV.nc", "-t", "VVL_y0001_gridT.nc", "-ntr", "-vvl", "-o", "ms_vvl.nc"], ["ms_vvl.nc"]),
    ("cdfmocsig_ntr_full", "cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-ntr", "-full", "-o", "ms_f.nc"], ["ms_f.nc"]),
    # the ten programs of SURVEY 8f
    ("cdfvhst_default", "cdfvhst", ["-f", "vt.nc", "-o", "vh.nc"], ["vh.nc"]),
    ("cdfvhst_full", "cdfvhst", ["-f", "vt.nc", "-full", "-o", "vhf.nc"], ["vhf.nc"]),
    ("cdfvtrp_default", "cdfvtrp", UV, ["trp.nc"]),
    ("cdfvtrp_full", "cdfvtrp", UV + ["-full", "-o", "trpf.nc"], ["trpf.nc"]),
    ("cdfmean_var_S", "cdfmean", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-var", "-S"],
     ["cdfmean.nc", "cdfmean.txt", "cdfvar.txt", "cdfsum.txt"]),
    ("cdfmean_window_basins_full", "cdfmean", ["-f", "SYN_y0001_gridT.nc", "-v", "vosaline", "-p", "T", "-w", "3", "20", "2", "15", "2", "4",
                                    "-B", "new_maskglo.nc", "tmaskatl", "tmaskpac", "-full", "-o", "mw.nc", "-P", "pre_"], ["pre_mw.nc", "pre_cdfmean.txt"]),
    ("cdfsum_default", "cdfsum", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "T"], ["cdfsum.nc"]),
    ("cdfsum_window_mask_nowght", "cdfsum", ["-f", "SYN_y0001_gridV.nc", "-v", "vomecrty", "-p", "V", "-w", "3", "20", "2", "15", "2", "4",
                                   "-M", "mask.nc", "umask", "-nowght", "-o", "w.nc"], ["w.nc"]),
    ("cdfzonalsum_basins", "cdfzonalsum", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-b", "new_maskglo.nc"], ["zonalsum.nc"]),
    ("cdfzonalsum_list_pdeg_pdep", "cdfzonalsum", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-l", "vosaline", "-pdeg", "-pdep", "-o", "zs2.nc"], ["zonalintdeg.nc"]),      # (-pdeg: the name the reference writes, whatever -o says)
    ("cdfzonalmean_basins", "cdfzonalmean", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-b", "new_maskglo.nc"], ["zonalmean.nc"]),
    ("cdfzonalmean_max_list", "cdfzonalmean", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-b", "new_maskglo.nc", "-max", "-l", "votemper", "-o", "zm2.nc"], ["zm2.nc"]),
    ("cdfvint_default", "cdfvint", ["-f", "SYN_y0001_gridT.nc"], ["voheatc.nc"]),
    ("cdfvint_OCCI_smean_full_window", "cdfvint", ["-f", "SYN_y0001_gridT.nc", "-v", "vosaline", "-OCCI", "-smean", "-full", "-w", "3", "20", "2", "15",
                                        "-o", "sm.nc"], ["sm.nc"]),
    ("cdfvertmean_default", "cdfvertmean", ["-f", "SYN_y0001_gridT.nc", "-l", "votemper,vosaline", "-p", "T", "-zlim", "15", "120"], ["vertmean.nc"]),
    ("cdfvertmean_reversed_limits", "cdfvertmean", ["-f", "SYN_y0001_gridT.nc", "-l", "votemper", "-p", "T", "-zlim", "100", "10", "-o", "vm2.nc"], []),     # (refused: STOP 99 and a message)
    # (cdfmocsig only with -ntr: the reference never initialises its lntr (src/cdfmocsig.f90:101) and this build of it takes the
    #  neutral-density branch whatever the options say; with -ntr both sides mean the same)
    ("cdfmocsig_ntr_isodep", "cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-ntr", "-isodep"], ["mocsig.nc"]),
    ("cdfmocsig_ntr", "cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-ntr", "-sigmin", "19", "-sigstp", "0.25", "-nbins", "60",
                       "-o", "ntr.nc"], ["ntr.nc"]),
    # fourth tier (round 5): the options no case had yet -- other file and variable names (-s, -b, -vtvar, -namU, -namV), -nometric,
    # cdfvT -vvl, -debug, -pdep / -ndep_in, a window of cdfvertmean, cdfmocsig -eiv / -teos10 / -verbose (with -ntr: see above)
    ("cdfmoc_decomp_sfile", "cdfmoc", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-s", "salinity_alt.nc", "-decomp", "-o", "moc_ds.nc"],
     ["moc_ds.nc"]),
    ("cdfcurl_nometric", "cdfcurl", ["-u", "SYN_y0001_gridU.nc", "vozocrtx", "-v", "SYN_y0001_gridV.nc", "vomecrty", "-l", "1", "-nometric"], ["curl_grid.nc"]),
    ("cdfmhst_sfile_basin_file", "cdfmhst", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-s", "salinity_alt.nc", "-b", "basins_alt.nc", "-MST",
                                  "-o", "mhst_b.nc"], ["mhst_b.nc", "zonal_heat_trp.dat", "zonal_salt_trp.dat"]),
    ("cdfmhst_vtvar", "cdfmhst", ["-vt", "ALT_vt.nc", "-vtvar", "vt_alt", "vs_alt", "-o", "mhst_a.nc"], ["mhst_a.nc", "zonal_heat_trp.dat", "zonal_salt_trp.dat"]),
    ("cdfvT_namU_namV", "cdfvT", ["-c", "ALT", "-l", "y0001", "-namU", "uoce", "-namV", "voce", "-o", "vt_alt.nc"], ["vt_alt.nc"]),
    ("cdfvT_vvl", "cdfvT", ["-c", "VVL", "-l", "y0001", "-vvl", "-o", "vt_vvl2.nc"], ["vt_vvl2.nc"]),
    ("cdfzonalmean_pdep_ndep_in_debug", "cdfzonalmean", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-pdep", "-ndep_in", "-debug", "-o", "zm3.nc"], ["zm3.nc"]),
    ("cdfzonalsum_debug", "cdfzonalsum", ["-f", "SYN_y0001_gridT.nc", "-p", "T", "-debug", "-o", "zs3.nc"], ["zs3.nc"]),
    ("cdfvertmean_window_debug", "cdfvertmean", ["-f", "SYN_y0001_gridT.nc", "-l", "votemper", "-p", "T", "-zlim", "15", "120", "-w", "3", "20", "2", "15",
                                  "-debug", "-o", "vm_w.nc"], ["vm_w.nc"]),
    ("cdfmocsig_ntr_eiv_sfile_verbose", "cdfmocsig", ["-v", "EIV_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-s", "salinity_alt.nc", "-ntr", "-eiv",
                                         "-verbose", "-o", "ms_e.nc"], ["ms_e.nc"]),
    ("cdfmocsig_ntr_teos10", "cdfmocsig", ["-v", "SYN_y0001_gridV.nc", "-t", "SYN_y0001_gridT.nc", "-ntr", "-teos10", "-o", "ms_t.nc"], ["ms_t.nc"]),
    # cdfmean's other options (round 5: the reference program runs once it is built with its own gfortran spelling of line 194,
    # oracle/ref_prog_build.sh)
    ("cdfmean_M_names", "cdfmean", ["-f", "SYN_y0001_gridV.nc", "-v", "vomecrty", "-p", "V", "-M", "mask.nc", "umask", "-var", "-S", "-o", "mm.nc", "-ot", "mm.txt",
                          "-ov", "mv.txt", "-os", "ms.txt"], ["mm.nc", "mm.txt", "mv.txt", "ms.txt"]),
    ("cdfmean_zeromean", "cdfmean", ["-f", "SYN_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-zeromean", "-oz", "zm0.nc", "-o", "mz.nc"],
     ["mz.nc", "zm0.nc", "cdfmean.txt"]),
    ("cdfmean_vvl", "cdfmean", ["-f", "VVL_y0001_gridT.nc", "-v", "votemper", "-p", "T", "-vvl", "-o", "mvvl.nc"], ["mvvl.nc", "cdfmean.txt"]),
    ("cdfmean_U_window_k", "cdfmean", ["-f", "SYN_y0001_gridU.nc", "-v", "vozocrtx", "-p", "U", "-w", "0", "0", "4", "12", "2", "5", "-o", "mu.nc"],
     ["mu.nc", "cdfmean.txt"]),
    # (depths of the window as the reference takes them: positive, metres -- it flips the sign of depthw when it reads moc.nc
    # (src/cdfmaxmoc.f90:123) and a window of negative depths leaves its ikmin / ikmax without a value: that, not the compiler, was
    # the segmentation fault that kept these two cases out of the differential in round 4)
    ("cdfmaxmoc_atl", "cdfmaxmoc", ["-f", "moc.nc", "-b", "atl", "-w", "-30", "60", "20", "5000"], ["maxmoc.nc"]),
    ("cdfmaxmoc_glo_other_window", "cdfmaxmoc", ["-f", "moc.nc", "-b", "glo", "-w", "-60", "20", "100", "3000", "-o", "maxmoc2.nc"], ["maxmoc2.nc"]),
]


# cdfmoc -rapid (src/cdfmoc.f90:598-1004) needs a grid on which cdf_findij finds its 26.5 N section: the files of
# tests/test_cli_rapid.make_rapid_case, in a directory of their own.  The reference ends the option with STOP 99 (:255) AFTER it has
# written rapid_moc.nc: exit status 99 with the file is the outcome to equal.
RAPID_CASES = [
    ("cdfmoc_rapid", "cdfmoc", ["-v", "V.nc", "-t", "T.nc", "-u", "U.nc", "-rapid"], ["rapid_moc.nc"]),
    ("cdfmoc_rapid_full_sfile", "cdfmoc", ["-v", "V.nc", "-t", "T.nc", "-s", "T.nc", "-u", "U.nc", "-rapid", "-full", "-o", "m.nc"], ["rapid_m.nc"]),   # (the name the reference derives, src/cdfmoc.f90:1202)
]

# the programs called without an argument: the usage text, the reference's own (src/cdfmoc.f90:130-210 ...), then this build's
# "MI355X BUILD" lines.  No GPU, no input file: compared on the CPU (tests/test_prog_diff.py::test_usage_texts...)
USAGE_CASES = [(p + "_usage", p, [], []) for p in
               ("cdfmoc", "cdfmhst", "cdfvT", "cdfheatc", "cdfpsi", "cdfcurl", "cdfsig0", "cdfsigi", "cdfvhst", "cdfvtrp", "cdfmean", "cdfsum",
                "cdfzonalsum", "cdfzonalmean", "cdfvint", "cdfvertmean", "cdfmocsig", "cdfmaxmoc")]


def fold_like_list_directed_output(text, width=79):
    """what flang's list-directed PRINT makes of records longer than `width` characters: cut at `width`, the rest on the next line
    behind one blank (the reference's usage is PRINT *: its own build folds "transport and th / e")"""
    out = []
    for ln in text.split("\n"):
        while len(ln) > width:
            out.append(ln[:width])
            ln = " " + ln[width:]
        out.append(ln)
    return "\n".join(out)


def extra_files(d, g, rec):
    """the files the second-tier cases need beside those of nemo_files.make_case: a mixed-layer depth and a sea-surface height"""
    import nemo_files
    mxl = (5 + 200 * np.abs(rec["t"][:, 0]) / 30).astype("float32")
    nemo_files.write_data(os.path.join(d, "mxl.nc"), g, "deptht", {"somxl010": mxl})
    rng = np.random.default_rng(77)
    npk, npj, npi = g["tmask"].shape
    ssh = (rng.standard_normal((rec["t"].shape[0], npj, npi)) * g["tmask"][0]).astype(np.float32)
    nemo_files.write_data(os.path.join(d, "SYN_y0001_grid2D.nc"), g, "deptht", {"sossheig": ssh})
    # third tier: files whose vertical metric differs from record to record (-vvl), as tests/test_cli_vvl.py writes them
    import lift_cases as lc
    f32 = np.float32
    e3 = {p: lc.e3_records(g[f"e3{p}_0"], rec["t"].shape[0], 8000 + i) for i, p in enumerate("tuv")}
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridV.nc"), g, "depthv", {"vomecrty": rec["v"], "e3v": e3["v"]})
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridU.nc"), g, "depthu", {"vozocrtx": rec["u"], "e3u": e3["u"]})
    nemo_files.write_data(os.path.join(d, "VVL_y0001_gridT.nc"), g, "deptht", {"votemper": rec["t"], "vosaline": rec["s"], "e3t": e3["t"]},
                          fill=0.0)
    vt = {"vomevt": (rec["t"] * rec["v"]).astype(f32), "vomevs": (rec["s"] * rec["v"]).astype(f32),
          "vozout": (rec["t"] * rec["u"]).astype(f32), "vozous": (rec["s"] * rec["u"]).astype(f32)}
    nemo_files.write_data(os.path.join(d, "VVL_vt.nc"), g, "deptht", dict(vt, e3u=e3["u"], e3v=e3["v"]))
    # fourth tier: the same fields under other file and variable names
    import shutil
    nemo_files.write_data(os.path.join(d, "salinity_alt.nc"), g, "deptht", {"vosaline": (rec["s"] * f32(0.97) * g["tmask"][None]).astype(f32)}, fill=0.0)
    shutil.copy(os.path.join(d, "new_maskglo.nc"), os.path.join(d, "basins_alt.nc"))
    nemo_files.write_data(os.path.join(d, "ALT_vt.nc"), g, "deptht", {"vt_alt": vt["vomevt"][:, ::-1].copy(), "vs_alt": vt["vomevs"][:, ::-1].copy()})
    nemo_files.write_data(os.path.join(d, "ALT_y0001_gridT.nc"), g, "deptht", {"votemper": rec["t"], "vosaline": rec["s"]}, fill=0.0)
    nemo_files.write_data(os.path.join(d, "ALT_y0001_gridU.nc"), g, "depthu", {"uoce": rec["u"][::-1].copy()})
    nemo_files.write_data(os.path.join(d, "ALT_y0001_gridV.nc"), g, "depthv", {"voce": rec["v"][::-1].copy()})
    veiv = (np.random.default_rng(79).standard_normal(rec["v"].shape).astype(f32) * f32(0.05) * g["vmask"][None]).astype(f32)
    nemo_files.write_data(os.path.join(d, "EIV_y0001_gridV.nc"), g, "depthv", {"vomecrty": rec["v"], "vomeeivv": veiv})


def _att(v):
    """an attribute value -> JSON-able [type, value]"""
    if isinstance(v, bytes):
        return ["char", v.decode("latin-1")]
    a = np.asarray(v)
    return [a.dtype.str.lstrip("<>=|"), a.ravel().tolist()]


def dump_nc(path):
    """-> (schema dict (JSON-able: dimensions, variables and attributes in file order), {variable: data})"""
    with netcdf_file(path, "r", mmap=False) as f:
        schema = {"version_byte": int(f.version_byte),
                  "dimensions": [[nm, (None if f.dimensions[nm] is None else int(f.dimensions[nm]))] for nm in f._dims],
                  "global": [[k, _att(v)] for k, v in f._attributes.items()],
                  "variables": []}
        data = {}
        for nm, v in f.variables.items():
            schema["variables"].append({"name": nm, "type": v.data.dtype.str.lstrip("<>=|"), "dims": list(v.dimensions),
                                        "shape": [int(x) for x in v.shape], "atts": [[k, _att(a)] for k, a in v._attributes.items()]})
            data[nm] = np.array(v[:]).astype(v.data.dtype.newbyteorder("="))
    return schema, data


def run_case(bindir, case, cwd, env=None):
    """run one case -> dict(rc, stdout, files={name: schema or text}), {file::var: array}"""
    name, prog, args, outs = case
    for o in outs:
        if os.path.exists(os.path.join(cwd, o)):
            os.remove(os.path.join(cwd, o))
    e = dict(os.environ)
    e.update(env or {})
    # argv[0] is the bare program name (it goes into the `history` attribute, src/modutils.f90 SetGlobalAtt): the same text
    # whichever directory the binary lies in
    # (a run that does not end fails at once, described: tests/run_program.py)
    p = run_program.run_watched([prog] + args, executable=os.path.join(bindir, prog), cwd=cwd, env=e, timeout=300)
    meta = {"case": name, "program": prog, "args": args, "rc": p.returncode, "stdout": p.stdout, "files": {}}
    if p.returncode != 0:
        meta["stderr"] = p.stderr
    arrays = {}
    for o in outs:
        path = os.path.join(cwd, o)
        if not os.path.exists(path):
            meta["files"][o] = None
        elif o.endswith(".nc"):
            schema, data = dump_nc(path)
            meta["files"][o] = schema
            for k, a in data.items():
                arrays[f"{o}::{k}"] = a
        else:
            meta["files"][o] = open(path).read()
    return meta, arrays


def save_golden(path, meta, arrays):
    np.savez_compressed(path, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)


def left_out():
    """the cases whose reference program does not run in the build container as it is (tests/golden/prog_left_out.json)"""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prog_left_out.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def load_golden(path):
    z = np.load(path)
    meta = json.loads(bytes(z["__meta__"]).decode())
    return meta, {k: z[k] for k in z.files if k != "__meta__"}
