/*
 * nch5.c -- read-only NetCDF-4 support: maps the HDF5 layout netCDF-4 uses onto the classic data model of
 * ncclassic.c (dimensions, variables, attributes), so that cdfio reads a NetCDF-4 file like a classic one.
 * Built as a plug-in (libnch5.so, linked against the HDF5 C library >= 1.10) that libncclassic.so loads when
 * it first meets an HDF5 signature; written from the published netCDF-4 format description ("NetCDF-4 Format"
 * appendix of the NetCDF User's Guide), no code from libnetcdf.
 *
 * What is understood (the root group only, which is what NEMO / XIOS write):
 *   - a dataset with CLASS = "DIMENSION_SCALE" is a dimension: its name, its extent, unlimited when its
 *     maximum extent is; `_Netcdf4Dimid` gives the dimension id when present.  A scale whose NAME starts with
 *     "This is a netCDF dimension but not a netCDF variable." is a dimension only, any other one is also the
 *     coordinate variable of that dimension;
 *   - every other dataset is a variable; its dimensions are the scales referenced by DIMENSION_LIST (object
 *     references), a dataset without them gets anonymous "phony_dim_<n>" dimensions by extent;
 *     the prefix "_nc4_non_coord_" of a variable named like a dimension is dropped;
 *   - integer and floating-point types of 1, 2, 4, 8 bytes; chunked / deflated / shuffled storage is HDF5's
 *     business; values are converted to the requested memory type by H5Dread;
 *   - attributes: numeric arrays, and strings (fixed or variable length) as NC_CHAR; the bookkeeping
 *     attributes of the format itself are hidden.
 * Not understood: groups, user-defined / compound / enum / opaque / vlen types, string variables.
 *
 * Output (nch5_create ...): the same layout written from a complete model -- dimension scales made by HDF5's
 * High-Level library (H5DSset_scale / H5DSattach_scale, as libnetcdf does), `_Netcdf4Dimid`, `_NCProperties`, creation
 * order tracked, chunked storage with optional shuffle + deflate, an unlimited record dimension that grows with the
 * data, fill value taken from `_FillValue`.
 */
#define _GNU_SOURCE
#include <hdf5.h>
#include <hdf5_hl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include "ncmodel.h"

#if H5_VERS_MAJOR == 1 && H5_VERS_MINOR < 10
#error "HDF5 1.10 or later is needed (hid_t is 64 bits, H5Rdereference2)"
#endif

/* chunks that went through the threaded routes since the library was loaded (tests check that the routes are taken) */
static unsigned long g_chunks_read, g_chunks_written;
unsigned long nch5_chunks_inflated(void) { return g_chunks_read; }
unsigned long nch5_chunks_deflated(void) { return g_chunks_written; }

/* threads that inflate / deflate chunks: CDFT_IO_THREADS, else half the cores between 8 and 64 (deflate of a 2 MB chunk of
 * floats is ~25 ms: a level of eORCA12 is 31 of them, a slab of the feed 62) */
static int filter_threads(void) {
    int n = 8;
    const long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (ncpu / 2 > n) n = ncpu / 2 > 64 ? 64 : (int)(ncpu / 2);
    const char *e = getenv("CDFT_IO_THREADS");
    if (e && atoi(e) >= 1) n = atoi(e);
    return n > 64 ? 64 : n;
}

/* what the chunk-at-a-time writer needs to know about an output variable */
typedef struct {
    int rank, type, deflate, shuffle;      /* deflate 0: the variable goes through H5Dwrite */
    size_t esz;
    hsize_t chunk[NCC_MAX_DIMS];
} VarOut;

struct NcH5 {
    hid_t file;
    int nvars;
    hid_t *dset;      /* one per variable */
    VarOut *vout;     /* output only, one per variable */
    /* output only */
    int ndims, unlim;
    hid_t *scale;     /* one per dimension: the coordinate variable's dataset or a dimension-only scale */
    int *own_scale;   /* 1 when the scale is a dimension-only dataset (closed here, extended at the end) */
};

typedef struct {
    char name[NCC_MAX_NAME];
    hid_t id;
    int rank;
    hsize_t ext[NCC_MAX_DIMS], maxext[NCC_MAX_DIMS];
    int is_scale, dim_only, dimid_att;     /* dimid_att: _Netcdf4Dimid or -1 */
    int dim_index;                         /* index of the dimension this scale defines */
} Obj;

typedef struct {
    Obj *o;
    int n, cap;
    hid_t file;
} Scan;

static const char *const kDimOnly = "This is a netCDF dimension but not a netCDF variable.";
static const char *const kNonCoord = "_nc4_non_coord_";

/* ---- small helpers ---------------------------------------------------------------------------------------- */
static int read_string_att(hid_t obj, const char *name, char *out, size_t cap) {
    if (H5Aexists(obj, name) <= 0) return -1;
    hid_t a = H5Aopen(obj, name, H5P_DEFAULT);
    if (a < 0) return -1;
    hid_t t = H5Aget_type(a);
    int rc = -1;
    out[0] = 0;
    if (H5Tget_class(t) == H5T_STRING) {
        hid_t mt = H5Tcopy(H5T_C_S1);
        if (H5Tis_variable_str(t) > 0) {
            char *p = NULL;
            H5Tset_size(mt, H5T_VARIABLE);
            if (H5Aread(a, mt, &p) >= 0 && p) { snprintf(out, cap, "%s", p); H5free_memory(p); rc = 0; }
        } else {
            const size_t n = H5Tget_size(t);
            char *buf = (char *)calloc(n + 1, 1);
            H5Tset_size(mt, n + 1);
            if (buf && H5Aread(a, mt, buf) >= 0) { snprintf(out, cap, "%s", buf); rc = 0; }
            free(buf);
        }
        H5Tclose(mt);
    }
    H5Tclose(t);
    H5Aclose(a);
    return rc;
}

static int nc_type_of(hid_t t) {
    const size_t sz = H5Tget_size(t);
    switch (H5Tget_class(t)) {
        case H5T_INTEGER: {
            const int sgn = H5Tget_sign(t) != H5T_SGN_NONE;
            if (sz == 1) return sgn ? NCC_BYTE : NCC_UBYTE;
            if (sz == 2) return sgn ? NCC_SHORT : NCC_USHORT;
            if (sz == 4) return sgn ? NCC_INT : NCC_UINT;
            if (sz == 8) return sgn ? NCC_INT64 : NCC_UINT64;
            return 0;
        }
        case H5T_FLOAT:
            return sz == 4 ? NCC_FLOAT : (sz == 8 ? NCC_DOUBLE : 0);
        default:
            return 0;
    }
}

static hid_t mem_type(int t) {
    switch (t) {
        case NCC_BYTE: return H5T_NATIVE_SCHAR;
        case NCC_UBYTE: return H5T_NATIVE_UCHAR;
        case NCC_SHORT: return H5T_NATIVE_SHORT;
        case NCC_USHORT: return H5T_NATIVE_USHORT;
        case NCC_INT: return H5T_NATIVE_INT;
        case NCC_UINT: return H5T_NATIVE_UINT;
        case NCC_INT64: return H5T_NATIVE_LLONG;
        case NCC_UINT64: return H5T_NATIVE_ULLONG;
        case NCC_FLOAT: return H5T_NATIVE_FLOAT;
        case NCC_DOUBLE: return H5T_NATIVE_DOUBLE;
        default: return -1;
    }
}

static size_t mem_size(int t) {
    switch (t) {
        case NCC_BYTE: case NCC_UBYTE: case NCC_CHAR: return 1;
        case NCC_SHORT: case NCC_USHORT: return 2;
        case NCC_INT: case NCC_UINT: case NCC_FLOAT: return 4;
        default: return 8;
    }
}

/* ---- attributes -------------------------------------------------------------------------------------------- */
typedef struct { Att *a; int n, cap; } AttList;

static int hidden_att(const char *name) {
    static const char *const k[] = {"DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates",
                                    "_NCProperties", "_nc3_strict", "_NetCDF4Dimid"};
    for (size_t i = 0; i < sizeof k / sizeof k[0]; ++i) if (!strcmp(name, k[i])) return 1;
    return 0;
}

static herr_t att_cb(hid_t obj, const char *name, const H5A_info_t *info, void *ctx) {
    (void)info;
    AttList *l = (AttList *)ctx;
    if (hidden_att(name)) return 0;
    hid_t a = H5Aopen(obj, name, H5P_DEFAULT);
    if (a < 0) return 0;
    hid_t t = H5Aget_type(a), sp = H5Aget_space(a);
    Att at;
    memset(&at, 0, sizeof at);
    snprintf(at.name, sizeof at.name, "%s", name);
    const hssize_t npts = H5Sget_simple_extent_npoints(sp);
    int keep = 0;
    if (H5Tget_class(t) == H5T_STRING) {
        char *text = NULL;
        hid_t mt = H5Tcopy(H5T_C_S1);
        if (H5Tis_variable_str(t) > 0) {
            char **p = (char **)calloc((size_t)(npts > 0 ? npts : 1), sizeof *p);
            H5Tset_size(mt, H5T_VARIABLE);
            if (p && H5Aread(a, mt, p) >= 0 && p[0]) text = strdup(p[0]);
            if (p) { for (hssize_t i = 0; i < npts; ++i) if (p[i]) H5free_memory(p[i]); free(p); }
        } else {
            const size_t n = H5Tget_size(t);
            char *buf = (char *)calloc((size_t)(npts > 0 ? npts : 1) * (n + 1) + 1, 1);
            H5Tset_size(mt, n + 1);
            if (buf && H5Aread(a, mt, buf) >= 0) text = buf; else free(buf);
        }
        H5Tclose(mt);
        if (text) {
            at.type = NCC_CHAR;
            at.nelems = strlen(text);
            at.data = text;
            keep = 1;
        }
    } else {
        at.type = nc_type_of(t);
        if (at.type && npts > 0) {
            at.nelems = (size_t)npts;
            at.data = malloc(at.nelems * mem_size(at.type));
            if (at.data && H5Aread(a, mem_type(at.type), at.data) >= 0) keep = 1;
            else free(at.data);
        }
    }
    if (keep) {
        if (l->n == l->cap) {
            l->cap = l->cap ? 2 * l->cap : 8;
            l->a = (Att *)realloc(l->a, (size_t)l->cap * sizeof(Att));
        }
        l->a[l->n++] = at;
    }
    H5Sclose(sp);
    H5Tclose(t);
    H5Aclose(a);
    return 0;
}

static void read_atts(hid_t obj, Att **out, int *n) {
    AttList l = {NULL, 0, 0};
    hsize_t idx = 0;
    if (H5Aiterate2(obj, H5_INDEX_CRT_ORDER, H5_ITER_INC, &idx, att_cb, &l) < 0) {      /* creation order not tracked */
        idx = 0;
        for (int i = 0; i < l.n; ++i) free(l.a[i].data);
        l.n = 0;
        H5Aiterate2(obj, H5_INDEX_NAME, H5_ITER_INC, &idx, att_cb, &l);
    }
    *out = l.a;
    *n = l.n;
}

/* ---- objects of the root group ----------------------------------------------------------------------------- */
static herr_t link_cb(hid_t grp, const char *name, const H5L_info_t *info, void *ctx) {
    (void)info;
    Scan *s = (Scan *)ctx;
    hid_t d = H5Dopen2(grp, name, H5P_DEFAULT);
    if (d < 0) return 0;                                         /* a group, a named type, ...: not ours */
    if (s->n == s->cap) {
        s->cap = s->cap ? 2 * s->cap : 32;
        s->o = (Obj *)realloc(s->o, (size_t)s->cap * sizeof(Obj));
    }
    Obj *o = &s->o[s->n];
    memset(o, 0, sizeof *o);
    snprintf(o->name, sizeof o->name, "%s", name);
    o->id = d;
    o->dimid_att = -1;
    o->dim_index = -1;
    hid_t sp = H5Dget_space(d);
    o->rank = H5Sget_simple_extent_ndims(sp);
    if (o->rank < 0 || o->rank > NCC_MAX_DIMS) { H5Sclose(sp); H5Dclose(d); return 0; }
    if (o->rank > 0) H5Sget_simple_extent_dims(sp, o->ext, o->maxext);
    H5Sclose(sp);
    char cls[64];
    if (read_string_att(d, "CLASS", cls, sizeof cls) == 0 && !strcmp(cls, "DIMENSION_SCALE")) {
        char nm[NCC_MAX_NAME];
        o->is_scale = 1;
        if (read_string_att(d, "NAME", nm, sizeof nm) == 0 && !strncmp(nm, kDimOnly, strlen(kDimOnly))) o->dim_only = 1;
        if (H5Aexists(d, "_Netcdf4Dimid") > 0) {
            hid_t a = H5Aopen(d, "_Netcdf4Dimid", H5P_DEFAULT);
            int v = -1;
            if (a >= 0 && H5Aread(a, H5T_NATIVE_INT, &v) >= 0) o->dimid_att = v;
            if (a >= 0) H5Aclose(a);
        }
    }
    ++s->n;
    return 0;
}

static int find_dim(const Dim *dims, int nd, const char *name) {
    for (int i = 0; i < nd; ++i) if (!strcmp(dims[i].name, name)) return i;
    return -1;
}

int nch5_open(const char *path, NcH5 **hout, int *ndims, Dim **dims_out, int *nvars, Var **vars_out, int *ngatts, Att **gatts,
              int *unlim, uint64_t *numrecs) {
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);                       /* failures are reported through return codes */
    hid_t file = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (file < 0) return NCC_EHDFERR;
    Scan s = {NULL, 0, 0, file};
    hsize_t idx = 0;
    if (H5Literate(file, H5_INDEX_CRT_ORDER, H5_ITER_INC, &idx, link_cb, &s) < 0) {     /* creation order not tracked */
        for (int i = 0; i < s.n; ++i) H5Dclose(s.o[i].id);
        s.n = 0; idx = 0;
        H5Literate(file, H5_INDEX_NAME, H5_ITER_INC, &idx, link_cb, &s);
    }
    /* dimensions: the scales, ordered by _Netcdf4Dimid when every scale has one */
    int nscale = 0, all_ids = 1;
    for (int i = 0; i < s.n; ++i) if (s.o[i].is_scale && s.o[i].rank >= 1) { ++nscale; if (s.o[i].dimid_att < 0) all_ids = 0; }
    int cap = nscale + s.n * NCC_MAX_DIMS + 1;
    Dim *dims = (Dim *)calloc((size_t)cap, sizeof(Dim));
    uint64_t *dlen = (uint64_t *)calloc((size_t)cap, sizeof(uint64_t));            /* current lengths */
    int nd = 0;
    *unlim = -1; *numrecs = 0;
    for (int pass = 0; pass < nscale; ++pass) {
        int pick = -1;
        for (int i = 0; i < s.n; ++i) {
            Obj *o = &s.o[i];
            if (!o->is_scale || o->rank < 1 || o->dim_index >= 0) continue;
            if (!all_ids) { pick = i; break; }
            if (pick < 0 || o->dimid_att < s.o[pick].dimid_att) pick = i;
        }
        if (pick < 0) break;
        Obj *o = &s.o[pick];
        o->dim_index = nd;
        snprintf(dims[nd].name, sizeof dims[nd].name, "%s", o->name);
        dlen[nd] = o->ext[0];
        const int is_unlim = o->maxext[0] == H5S_UNLIMITED;
        if (is_unlim && *unlim < 0) { *unlim = nd; *numrecs = o->ext[0]; dims[nd].len = 0; }
        else dims[nd].len = o->ext[0];
        ++nd;
    }
    /* variables */
    const size_t nobj = s.n > 0 ? (size_t)s.n : 1;
    Var *vars = (Var *)calloc(nobj, sizeof(Var));
    hid_t *dset = (hid_t *)calloc(nobj, sizeof(hid_t));
    int nv = 0, nphony = 0;
    for (int i = 0; i < s.n; ++i) {
        Obj *o = &s.o[i];
        if (o->is_scale && o->dim_only) { H5Dclose(o->id); continue; }
        hid_t t = H5Dget_type(o->id);
        const int xtype = nc_type_of(t);
        H5Tclose(t);
        if (!xtype) { H5Dclose(o->id); continue; }               /* strings, compounds, ...: not carried */
        Var *v = &vars[nv];
        const char *nm = o->name;
        if (!strncmp(nm, kNonCoord, strlen(kNonCoord))) nm += strlen(kNonCoord);
        snprintf(v->name, sizeof v->name, "%s", nm);
        v->type = xtype;
        v->ndims = o->rank;
        int have = 0;
        if (o->rank > 0 && H5Aexists(o->id, "DIMENSION_LIST") > 0) {
            hid_t a = H5Aopen(o->id, "DIMENSION_LIST", H5P_DEFAULT);
            hid_t at = H5Aget_type(a), asp = H5Aget_space(a);
            hvl_t *vl = (hvl_t *)calloc((size_t)o->rank, sizeof *vl);
            if (H5Sget_simple_extent_npoints(asp) == o->rank && H5Aread(a, at, vl) >= 0) {
                have = 1;
                for (int r = 0; r < o->rank; ++r) {
                    int di = -1;
                    if (vl[r].len >= 1 && vl[r].p) {
                        hobj_ref_t ref = ((hobj_ref_t *)vl[r].p)[0];
                        hid_t so = H5Rdereference2(o->id, H5P_DEFAULT, H5R_OBJECT, &ref);
                        if (so >= 0) {
                            char full[NCC_MAX_NAME + 2];
                            if (H5Iget_name(so, full, sizeof full) > 0) {
                                const char *base = strrchr(full, '/');
                                di = find_dim(dims, nd, base ? base + 1 : full);
                            }
                            H5Oclose(so);
                        }
                    }
                    if (di < 0) have = 0;
                    v->dimids[r] = di;
                }
                H5Dvlen_reclaim(at, asp, H5P_DEFAULT, vl);
            }
            free(vl);
            H5Sclose(asp); H5Tclose(at); H5Aclose(a);
        }
        if (!have && o->is_scale && o->rank == 1 && o->dim_index >= 0) { v->dimids[0] = o->dim_index; have = 1; }
        if (!have && o->rank > 0) {                              /* plain HDF5 dataset: anonymous dimensions by extent */
            for (int r = 0; r < o->rank; ++r) {
                int di = -1;
                for (int k = 0; k < nd; ++k)
                    if (!strncmp(dims[k].name, "phony_dim_", 10) && dlen[k] == o->ext[r]) { di = k; break; }
                if (di < 0) {
                    di = nd++;
                    snprintf(dims[di].name, sizeof dims[di].name, "phony_dim_%d", nphony++);
                    dims[di].len = dlen[di] = o->ext[r];
                }
                v->dimids[r] = di;
            }
        }
        v->isrec = (o->rank > 0 && v->dimids[0] == *unlim);
        read_atts(o->id, &v->atts, &v->natts);
        dset[nv] = o->id;
        ++nv;
    }
    hid_t root = H5Gopen2(file, "/", H5P_DEFAULT);
    *gatts = NULL; *ngatts = 0;
    if (root >= 0) { read_atts(root, gatts, ngatts); H5Gclose(root); }
    free(s.o);
    free(dlen);
    NcH5 *h = (NcH5 *)calloc(1, sizeof *h);
    h->file = file; h->nvars = nv; h->dset = dset;
    *hout = h; *ndims = nd; *dims_out = dims; *nvars = nv; *vars_out = vars;
    return 0;
}

/* ---- deflated variables, chunk at a time ---------------------------------------------------------------------
 * H5Dread inflates every chunk on the calling thread.  When the variable is chunked with no filter other than
 * shuffle and deflate and is stored in the type asked for, the chunks a request touches are fetched raw
 * (H5Dread_chunk, one thread: the library is not thread-safe) and inflated, unshuffled and copied into place by
 * CDFT_IO_THREADS threads (default 8).  Anything else -- other filters, a type conversion, a chunk that was never
 * written -- takes the library's route.  CDFT_NC4_SERIAL=1 forces that route (tests compare the two). */
typedef struct {
    int rank, shuffle, deflate;
    size_t esz, chunk_bytes;
    const hsize_t *chunk, *st, *ct;        /* chunk extents; the request */
    hsize_t c0[NCC_MAX_DIMS], nch[NCC_MAX_DIMS];
    size_t nchunks;
    void **raw; size_t *raw_bytes; uint32_t *mask;
    char *data;
    size_t next; int failed;
    pthread_mutex_t lock;
} ReadJob;

static void unshuffle_bytes(const unsigned char *in, unsigned char *out, size_t nbytes, size_t esz) {
    const size_t n = nbytes / esz;
    if (esz == 4) {
        for (size_t i = 0; i < n; ++i) {
            out[4 * i] = in[i]; out[4 * i + 1] = in[n + i]; out[4 * i + 2] = in[2 * n + i]; out[4 * i + 3] = in[3 * n + i];
        }
    } else {
        for (size_t b = 0; b < esz; ++b)
            for (size_t i = 0; i < n; ++i) out[i * esz + b] = in[b * n + i];
    }
    memcpy(out + n * esz, in + n * esz, nbytes - n * esz);
}

static void *read_worker(void *arg) {
    ReadJob *j = (ReadJob *)arg;
    const int nd = j->rank;
    unsigned char *a = (unsigned char *)malloc(j->chunk_bytes), *b = (unsigned char *)malloc(j->chunk_bytes);
    for (;;) {
        pthread_mutex_lock(&j->lock);
        const size_t c = j->next++;
        pthread_mutex_unlock(&j->lock);
        if (c >= j->nchunks || !a || !b) break;
        const unsigned char *cur = (const unsigned char *)j->raw[c];
        int bad = 0;
        if (j->deflate && !(j->mask[c] & (j->shuffle ? 2u : 1u))) {          /* deflate is the last filter of the pipeline */
            uLongf n = (uLongf)j->chunk_bytes;
            if (uncompress(a, &n, cur, (uLong)j->raw_bytes[c]) != Z_OK || n != j->chunk_bytes) bad = 1;
            cur = a;
        } else if (j->raw_bytes[c] != j->chunk_bytes) bad = 1;
        if (!bad && j->shuffle && j->esz > 1 && !(j->mask[c] & 1u)) { unshuffle_bytes(cur, b, j->chunk_bytes, j->esz); cur = b; }
        if (bad) { pthread_mutex_lock(&j->lock); j->failed = 1; pthread_mutex_unlock(&j->lock); break; }
        /* the part of the chunk inside the request, row by row */
        hsize_t org[NCC_MAX_DIMS], lo[NCC_MAX_DIMS], hi[NCC_MAX_DIMS];
        size_t rem = c, nrows = 1;
        for (int r = nd - 1; r >= 0; --r) {
            org[r] = (j->c0[r] + rem % j->nch[r]) * j->chunk[r];
            rem /= j->nch[r];
            lo[r] = org[r] > j->st[r] ? org[r] : j->st[r];
            hi[r] = org[r] + j->chunk[r] < j->st[r] + j->ct[r] ? org[r] + j->chunk[r] : j->st[r] + j->ct[r];
        }
        for (int r = 0; r < nd - 1; ++r) nrows *= hi[r] - lo[r];
        hsize_t idx[NCC_MAX_DIMS];
        for (int r = 0; r < nd; ++r) idx[r] = lo[r];
        for (size_t row = 0; row < nrows; ++row) {
            size_t src = 0, dst = 0;
            for (int r = 0; r < nd; ++r) {
                src = src * j->chunk[r] + (idx[r] - org[r]);
                dst = dst * j->ct[r] + (idx[r] - j->st[r]);
            }
            memcpy(j->data + dst * j->esz, cur + src * j->esz, (hi[nd - 1] - lo[nd - 1]) * j->esz);
            for (int r = nd - 2; r >= 0; --r) { if (++idx[r] < hi[r]) break; idx[r] = lo[r]; }
        }
    }
    free(a); free(b);
    return NULL;
}

/* 1: read; 0: not eligible or not possible this way (the caller uses H5Dread) */
static int get_by_chunks(hid_t d, int nd, const hsize_t *st, const hsize_t *ct, int memtype, void *data) {
    const char *off = getenv("CDFT_NC4_SERIAL");
    if (off && off[0] == '1') return 0;
    if (nd < 1) return 0;
    int ok = 0;
    hsize_t chunk[NCC_MAX_DIMS];
    ReadJob j;
    memset(&j, 0, sizeof j);
    hid_t dcpl = H5Dget_create_plist(d), ft = H5Dget_type(d);
    if (dcpl >= 0 && ft >= 0 && H5Pget_layout(dcpl) == H5D_CHUNKED && H5Pget_chunk(dcpl, nd, chunk) == nd &&
        H5Tequal(ft, mem_type(memtype)) > 0) {
        const int nf = H5Pget_nfilters(dcpl);
        ok = nf >= 1 && nf <= 2;
        for (int f = 0; f < nf && ok; ++f) {
            unsigned flags, cfg;
            size_t ncd = 8;
            unsigned cd[8];
            char name[8];
            const H5Z_filter_t id = H5Pget_filter2(dcpl, (unsigned)f, &flags, &ncd, cd, sizeof name, name, &cfg);
            if (id == H5Z_FILTER_SHUFFLE && f == 0 && nf == 2) j.shuffle = 1;
            else if (id == H5Z_FILTER_DEFLATE && f == nf - 1) j.deflate = 1;
            else ok = 0;
        }
        if (ok && !j.deflate) ok = 0;
    }
    if (ft >= 0) H5Tclose(ft);
    if (dcpl >= 0) H5Pclose(dcpl);
    if (!ok) return 0;
    j.rank = nd; j.esz = mem_size(memtype); j.chunk = chunk; j.st = st; j.ct = ct; j.data = (char *)data;
    j.nchunks = 1; j.chunk_bytes = j.esz;
    for (int r = 0; r < nd; ++r) {
        if (ct[r] == 0) return 0;
        j.c0[r] = st[r] / chunk[r];
        j.nch[r] = (st[r] + ct[r] - 1) / chunk[r] - j.c0[r] + 1;
        j.nchunks *= j.nch[r];
        j.chunk_bytes *= chunk[r];
    }
    if (j.nchunks < 2 || j.chunk_bytes > ((size_t)1 << 31)) return 0;      /* one chunk: nothing to share out */
    j.raw = (void **)calloc(j.nchunks, sizeof *j.raw);
    j.raw_bytes = (size_t *)calloc(j.nchunks, sizeof *j.raw_bytes);
    j.mask = (uint32_t *)calloc(j.nchunks, sizeof *j.mask);
    ok = j.raw && j.raw_bytes && j.mask;
    for (size_t c = 0; c < j.nchunks && ok; ++c) {
        hsize_t offset[NCC_MAX_DIMS], nbytes = 0;
        size_t rem = c;
        for (int r = nd - 1; r >= 0; --r) { offset[r] = (j.c0[r] + rem % j.nch[r]) * chunk[r]; rem /= j.nch[r]; }
        if (H5Dget_chunk_storage_size(d, offset, &nbytes) < 0 || nbytes == 0) { ok = 0; break; }   /* never written: fill values */
        j.raw[c] = malloc((size_t)nbytes);
        j.raw_bytes[c] = (size_t)nbytes;
        if (!j.raw[c] || H5Dread_chunk(d, H5P_DEFAULT, offset, &j.mask[c], j.raw[c]) < 0) ok = 0;
    }
    if (ok) {
        pthread_mutex_init(&j.lock, NULL);
        int nthreads = filter_threads();
        if ((size_t)nthreads > j.nchunks) nthreads = (int)j.nchunks;
        pthread_t th[64];
        int started = 0;
        for (int t = 1; t < nthreads; ++t)
            if (pthread_create(&th[started], NULL, read_worker, &j) == 0) ++started;
        read_worker(&j);
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
        pthread_mutex_destroy(&j.lock);
        if (j.failed) ok = 0;
    }
    if (j.raw) for (size_t c = 0; c < j.nchunks; ++c) free(j.raw[c]);
    free(j.raw); free(j.raw_bytes); free(j.mask);
    if (ok) g_chunks_read += j.nchunks;
    return ok ? 1 : 0;
}

int nch5_get_vara(NcH5 *h, int varid, int nd, const uint64_t *start, const uint64_t *count, int memtype, void *data) {
    if (!h || varid < 0 || varid >= h->nvars) return NCC_ENOTVAR;
    const hid_t mt = mem_type(memtype);
    if (mt < 0) return NCC_EBADTYPE;
    const hid_t d = h->dset[varid];
    hid_t fs = H5Dget_space(d);
    if (fs < 0) return NCC_EHDFERR;
    int rc = 0;
    if (H5Sget_simple_extent_ndims(fs) != nd) rc = NCC_EINVALCOORDS;
    if (!rc && nd == 0) {
        if (H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) rc = NCC_EHDFERR;
    } else if (!rc) {
        hsize_t st[NCC_MAX_DIMS], ct[NCC_MAX_DIMS], ext[NCC_MAX_DIMS];
        H5Sget_simple_extent_dims(fs, ext, NULL);
        for (int i = 0; i < nd; ++i) {
            st[i] = start[i]; ct[i] = count[i];
            if (st[i] + ct[i] > ext[i]) rc = NCC_EEDGE;          /* a variable shorter than its unlimited dimension */
        }
        if (!rc && get_by_chunks(d, nd, st, ct, memtype, data) == 1) { H5Sclose(fs); return 0; }
        if (!rc) {
            hid_t ms = H5Screate_simple(nd, ct, NULL);
            if (ms < 0 || H5Sselect_hyperslab(fs, H5S_SELECT_SET, st, NULL, ct, NULL) < 0 ||
                H5Dread(d, mt, ms, fs, H5P_DEFAULT, data) < 0)
                rc = NCC_EHDFERR;
            if (ms >= 0) H5Sclose(ms);
        }
    }
    H5Sclose(fs);
    return rc;
}

void nch5_close(NcH5 *h) {
    if (!h) return;
    for (int i = 0; i < h->nvars; ++i) H5Dclose(h->dset[i]);
    H5Fclose(h->file);
    free(h->dset);
    free(h);
}

int nch5_finish(NcH5 *h, uint64_t numrecs);

/* ---- output ------------------------------------------------------------------------------------------------ */
static hid_t file_type(int t) {
    switch (t) {
        case NCC_BYTE: return H5T_STD_I8LE;
        case NCC_UBYTE: return H5T_STD_U8LE;
        case NCC_SHORT: return H5T_STD_I16LE;
        case NCC_USHORT: return H5T_STD_U16LE;
        case NCC_INT: return H5T_STD_I32LE;
        case NCC_UINT: return H5T_STD_U32LE;
        case NCC_INT64: return H5T_STD_I64LE;
        case NCC_UINT64: return H5T_STD_U64LE;
        case NCC_FLOAT: return H5T_IEEE_F32LE;
        case NCC_DOUBLE: return H5T_IEEE_F64LE;
        default: return -1;
    }
}

static int write_att(hid_t obj, const Att *a) {
    if (H5Aexists(obj, a->name) > 0) H5Adelete(obj, a->name);
    hid_t sp, id;
    int rc = 0;
    if (a->type == NCC_CHAR) {                             /* as libnetcdf: a fixed-length string, scalar space */
        hid_t at = H5Tcopy(H5T_C_S1);
        H5Tset_size(at, a->nelems ? a->nelems : 1);
        H5Tset_strpad(at, H5T_STR_NULLTERM);
        sp = H5Screate(H5S_SCALAR);
        id = H5Acreate2(obj, a->name, at, sp, H5P_DEFAULT, H5P_DEFAULT);
        if (id < 0 || H5Awrite(id, at, a->nelems ? a->data : "") < 0) rc = NCC_EHDFERR;
        H5Tclose(at);
    } else {
        hsize_t n = a->nelems;
        if (file_type(a->type) < 0 || n == 0) return 0;
        sp = H5Screate_simple(1, &n, NULL);
        id = H5Acreate2(obj, a->name, file_type(a->type), sp, H5P_DEFAULT, H5P_DEFAULT);
        if (id < 0 || H5Awrite(id, mem_type(a->type), a->data) < 0) rc = NCC_EHDFERR;
    }
    if (id >= 0) H5Aclose(id);
    H5Sclose(sp);
    return rc;
}

int nch5_put_atts(NcH5 *h, int varid, int natts, const Att *atts) {
    if (!h) return NCC_EBADID;
    hid_t obj, root = -1;
    if (varid == NCC_GLOBAL) obj = root = H5Gopen2(h->file, "/", H5P_DEFAULT);
    else if (varid >= 0 && varid < h->nvars) obj = h->dset[varid];
    else return NCC_ENOTVAR;
    int rc = 0;
    for (int a = 0; a < natts && !rc; ++a) rc = write_att(obj, &atts[a]);
    if (root >= 0) H5Gclose(root);
    return rc;
}

int nch5_create(const char *path, NcH5 **hout, int ndims, const Dim *dims, int nvars, const Var *vars, int ngatts, const Att *gatts,
                int unlim) {
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t fcpl = H5Pcreate(H5P_FILE_CREATE);
    H5Pset_link_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    H5Pset_attr_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    hid_t file = H5Fcreate(path, H5F_ACC_TRUNC, fcpl, H5P_DEFAULT);
    H5Pclose(fcpl);
    if (file < 0) return NCC_EHDFERR;
    NcH5 *h = (NcH5 *)calloc(1, sizeof *h);
    h->file = file; h->nvars = nvars; h->ndims = ndims; h->unlim = unlim;
    h->dset = (hid_t *)calloc((size_t)(nvars > 0 ? nvars : 1), sizeof(hid_t));
    h->vout = (VarOut *)calloc((size_t)(nvars > 0 ? nvars : 1), sizeof(VarOut));
    h->scale = (hid_t *)calloc((size_t)(ndims > 0 ? ndims : 1), sizeof(hid_t));
    h->own_scale = (int *)calloc((size_t)(ndims > 0 ? ndims : 1), sizeof(int));
    int rc = 0;
    {                                                      /* what libnetcdf stamps on every file */
        char props[128];
        unsigned maj, min, rel;
        H5get_libversion(&maj, &min, &rel);
        snprintf(props, sizeof props, "version=2,cdftools_amd=1,hdf5=%u.%u.%u", maj, min, rel);
        Att a;
        memset(&a, 0, sizeof a);
        strcpy(a.name, "_NCProperties"); a.type = NCC_CHAR; a.nelems = strlen(props); a.data = props;
        hid_t root = H5Gopen2(file, "/", H5P_DEFAULT);
        rc = write_att(root, &a);
        for (int i = 0; i < ngatts && !rc; ++i) rc = write_att(root, &gatts[i]);
        H5Gclose(root);
    }
    /* the coordinate variable of a dimension: 1-D, same name, on that dimension */
    int *coord = (int *)calloc((size_t)(ndims > 0 ? ndims : 1), sizeof(int));
    for (int d = 0; d < ndims; ++d) {
        coord[d] = -1;
        h->scale[d] = -1;
        for (int v = 0; v < nvars; ++v)
            if (vars[v].ndims == 1 && vars[v].dimids[0] == d && !strcmp(vars[v].name, dims[d].name)) coord[d] = v;
    }
    for (int v = 0; v < nvars && !rc; ++v) {
        const Var *x = &vars[v];
        char h5name[NCC_MAX_NAME + 16];
        int clash = 0;
        for (int d = 0; d < ndims; ++d) if (!strcmp(x->name, dims[d].name) && coord[d] != v) clash = 1;
        snprintf(h5name, sizeof h5name, "%s%s", clash ? kNonCoord : "", x->name);
        hsize_t ext[NCC_MAX_DIMS], mx[NCC_MAX_DIMS], chunk[NCC_MAX_DIMS];
        int has_unlim = 0;
        for (int r = 0; r < x->ndims; ++r) {
            const int d = x->dimids[r];
            ext[r] = dims[d].len; mx[r] = ext[r];
            if (d == unlim) { ext[r] = 0; mx[r] = H5S_UNLIMITED; has_unlim = 1; }
            hsize_t c = x->chunk[r];
            if (d == unlim) c = (c >= 1 && c <= 1024) ? c : 1;
            else if (c < 1 || c > dims[d].len) c = dims[d].len ? dims[d].len : 1;      /* unset or out of range: whole dimension */
            chunk[r] = c;
        }
        if (x->ndims >= 3 && !(x->chunk[x->ndims - 1] >= 1))        /* nothing asked for: one horizontal slab per chunk */
            for (int r = 0; r < x->ndims - 2; ++r) chunk[r] = 1;
        hid_t sp = x->ndims ? H5Screate_simple(x->ndims, ext, mx) : H5Screate(H5S_SCALAR);
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        H5Pset_attr_creation_order(dcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
        if (x->ndims && (has_unlim || x->deflate > 0 || x->chunk[x->ndims - 1] >= 1)) {
            H5Pset_chunk(dcpl, x->ndims, chunk);
            if (x->deflate > 0) {
                if (x->shuffle) H5Pset_shuffle(dcpl);
                H5Pset_deflate(dcpl, (unsigned)x->deflate);
                VarOut *vo = &h->vout[v];
                vo->rank = x->ndims; vo->type = x->type; vo->deflate = x->deflate; vo->shuffle = x->shuffle ? 1 : 0;
                vo->esz = mem_size(x->type);
                for (int r = 0; r < x->ndims; ++r) vo->chunk[r] = chunk[r];
            }
        }
        for (int a = 0; a < x->natts; ++a)                  /* never-written cells read back as the fill value */
            if (!strcmp(x->atts[a].name, "_FillValue") && x->atts[a].type == x->type && x->atts[a].nelems == 1)
                H5Pset_fill_value(dcpl, mem_type(x->type), x->atts[a].data);
        h->dset[v] = H5Dcreate2(file, h5name, file_type(x->type), sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        H5Pclose(dcpl);
        H5Sclose(sp);
        if (h->dset[v] < 0) { rc = NCC_EHDFERR; break; }
        for (int a = 0; a < x->natts && !rc; ++a) rc = write_att(h->dset[v], &x->atts[a]);
        for (int d = 0; d < ndims; ++d)
            if (coord[d] == v) { h->scale[d] = h->dset[v]; if (H5DSset_scale(h->dset[v], NULL) < 0) rc = NCC_EHDFERR; }
    }
    for (int d = 0; d < ndims && !rc; ++d) {               /* dimensions without a coordinate variable */
        if (h->scale[d] >= 0) continue;
        hsize_t ext = (d == unlim) ? 0 : dims[d].len, mx = (d == unlim) ? H5S_UNLIMITED : ext, chunk = 512;
        hid_t sp = H5Screate_simple(1, &ext, &mx);
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        if (d == unlim) H5Pset_chunk(dcpl, 1, &chunk);
        h->scale[d] = H5Dcreate2(file, dims[d].name, H5T_IEEE_F32BE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        H5Pclose(dcpl);
        H5Sclose(sp);
        h->own_scale[d] = 1;
        char nm[128];
        snprintf(nm, sizeof nm, "%s%10d", kDimOnly, (int)dims[d].len);
        if (h->scale[d] < 0 || H5DSset_scale(h->scale[d], nm) < 0) rc = NCC_EHDFERR;
    }
    for (int d = 0; d < ndims && !rc; ++d) {
        Att a;
        memset(&a, 0, sizeof a);
        strcpy(a.name, "_Netcdf4Dimid"); a.type = NCC_INT; a.nelems = 1; a.data = &d;
        rc = write_att(h->scale[d], &a);
    }
    for (int v = 0; v < nvars && !rc; ++v) {
        int is_coord = 0;
        for (int d = 0; d < ndims; ++d) if (coord[d] == v) is_coord = 1;
        if (is_coord) continue;
        for (int r = 0; r < vars[v].ndims; ++r)
            if (H5DSattach_scale(h->dset[v], h->scale[vars[v].dimids[r]], (unsigned)r) < 0) rc = NCC_EHDFERR;
    }
    free(coord);
    if (rc) { nch5_finish(h, 0); return rc; }
    *hout = h;
    return 0;
}

/* ---- deflated variables, whole chunks at a time ----------------------------------------------------------------
 * H5Dwrite runs the shuffle + deflate filters of every chunk on the calling thread (about 80 MB/s: the -nc4 output
 * of cdfsigi at eORCA025 took 6 s).  When a write covers whole chunks and needs no type conversion, the chunks are
 * shuffled and deflated here by CDFT_IO_THREADS threads (default: half the cores, 8 to 64: each holds one chunk buffer of its own, so a 128-core host runs 64 threads on 64 chunks at a time) and handed over with H5Dwrite_chunk: the bytes in
 * the file are what the library's own filters would have produced, any NetCDF-4 reader inflates them. */
typedef struct {
    const VarOut *vo;
    const char *data;                      /* the caller's array, C order, shape count[] */
    hsize_t start[NCC_MAX_DIMS], count[NCC_MAX_DIMS], nch[NCC_MAX_DIMS];
    size_t nchunks, chunk_bytes;
    void **out; size_t *out_bytes;         /* per chunk */
    size_t next; int failed;
    pthread_mutex_t lock;
} ChunkJob;

static void shuffle_bytes(const unsigned char *in, unsigned char *out, size_t nbytes, size_t esz) {
    const size_t n = nbytes / esz;
    if (esz == 4) {
        for (size_t i = 0; i < n; ++i) {
            out[i] = in[4 * i]; out[n + i] = in[4 * i + 1]; out[2 * n + i] = in[4 * i + 2]; out[3 * n + i] = in[4 * i + 3];
        }
    } else {
        for (size_t b = 0; b < esz; ++b)
            for (size_t i = 0; i < n; ++i) out[b * n + i] = in[i * esz + b];
    }
    memcpy(out + n * esz, in + n * esz, nbytes - n * esz);
}

static void *chunk_worker(void *arg) {
    ChunkJob *j = (ChunkJob *)arg;
    const VarOut *vo = j->vo;
    const int nd = vo->rank;
    unsigned char *raw = (unsigned char *)malloc(j->chunk_bytes), *shf = (unsigned char *)malloc(j->chunk_bytes);
    for (;;) {
        pthread_mutex_lock(&j->lock);
        const size_t c = j->next++;
        pthread_mutex_unlock(&j->lock);
        if (c >= j->nchunks || !raw || !shf) break;
        /* chunk index -> its first element in the caller's array and the part of it inside the array */
        hsize_t first[NCC_MAX_DIMS], valid[NCC_MAX_DIMS];
        size_t rem = c;
        for (int r = nd - 1; r >= 0; --r) {
            const hsize_t q = rem % j->nch[r];
            rem /= j->nch[r];
            first[r] = q * vo->chunk[r];
            valid[r] = j->count[r] - first[r] < vo->chunk[r] ? j->count[r] - first[r] : vo->chunk[r];
        }
        int full = 1;
        for (int r = 0; r < nd; ++r) if (valid[r] != vo->chunk[r]) full = 0;
        if (!full) memset(raw, 0, j->chunk_bytes);           /* an edge chunk: the cells outside the variable */
        /* rows of valid[nd-1] elements */
        size_t nrows = 1;
        for (int r = 0; r < nd - 1; ++r) nrows *= valid[r];
        hsize_t idx[NCC_MAX_DIMS] = {0};
        for (size_t row = 0; row < nrows; ++row) {
            size_t src = 0, dst = 0;
            for (int r = 0; r < nd; ++r) {
                src = src * j->count[r] + (first[r] + (r < nd - 1 ? idx[r] : 0));
                dst = dst * vo->chunk[r] + (r < nd - 1 ? idx[r] : 0);
            }
            memcpy(raw + dst * vo->esz, j->data + src * vo->esz, valid[nd - 1] * vo->esz);
            for (int r = nd - 2; r >= 0; --r) { if (++idx[r] < valid[r]) break; idx[r] = 0; }
        }
        const unsigned char *in = raw;
        if (vo->shuffle && vo->esz > 1) { shuffle_bytes(raw, shf, j->chunk_bytes, vo->esz); in = shf; }
        uLongf zn = compressBound((uLong)j->chunk_bytes);
        void *z = malloc(zn);
        if (!z || compress2((Bytef *)z, &zn, in, (uLong)j->chunk_bytes, vo->deflate) != Z_OK) {
            free(z);
            pthread_mutex_lock(&j->lock); j->failed = 1; pthread_mutex_unlock(&j->lock);
            break;
        }
        j->out[c] = z; j->out_bytes[c] = zn;
    }
    free(raw); free(shf);
    return NULL;
}

/* 1: written; 0: not eligible (the caller falls back to H5Dwrite); < 0: error */
static int put_whole_chunks(NcH5 *h, int varid, int nd, const hsize_t *st, const hsize_t *ct, const hsize_t *ext, int memtype,
                            const void *data) {
    const VarOut *vo = &h->vout[varid];
    if (vo->deflate <= 0 || vo->rank != nd || memtype != vo->type || nd < 1) return 0;
    const char *off = getenv("CDFT_NC4_SERIAL");               /* A/B and test hook: the library's own filter pipeline */
    if (off && off[0] == '1') return 0;
    ChunkJob j;
    memset(&j, 0, sizeof j);
    j.vo = vo; j.data = (const char *)data; j.nchunks = 1; j.chunk_bytes = vo->esz;
    for (int r = 0; r < nd; ++r) {
        if (ct[r] == 0 || st[r] % vo->chunk[r]) return 0;
        if (ct[r] % vo->chunk[r] && st[r] + ct[r] != ext[r]) return 0;      /* a partial chunk only at the edge */
        j.start[r] = st[r]; j.count[r] = ct[r];
        j.nch[r] = (ct[r] + vo->chunk[r] - 1) / vo->chunk[r];
        j.nchunks *= j.nch[r];
        j.chunk_bytes *= vo->chunk[r];
    }
    if (j.chunk_bytes > ((size_t)1 << 31)) return 0;
    j.out = (void **)calloc(j.nchunks, sizeof *j.out);
    j.out_bytes = (size_t *)calloc(j.nchunks, sizeof *j.out_bytes);
    if (!j.out || !j.out_bytes) { free(j.out); free(j.out_bytes); return NCC_ENOMEM; }
    pthread_mutex_init(&j.lock, NULL);
    int nthreads = filter_threads();
    if ((size_t)nthreads > j.nchunks) nthreads = (int)j.nchunks;
    pthread_t th[64];
    int started = 0;
    for (int t = 1; t < nthreads; ++t)
        if (pthread_create(&th[started], NULL, chunk_worker, &j) == 0) ++started;
    chunk_worker(&j);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&j.lock);
    int rc = j.failed ? NCC_ENOMEM : 1;
    for (size_t c = 0; c < j.nchunks; ++c) {
        if (rc == 1 && !j.out[c]) rc = NCC_ENOMEM;
        if (rc == 1) {
            hsize_t offset[NCC_MAX_DIMS];
            size_t rem = c;
            for (int r = nd - 1; r >= 0; --r) { offset[r] = st[r] + (rem % j.nch[r]) * vo->chunk[r]; rem /= j.nch[r]; }
            if (H5Dwrite_chunk(h->dset[varid], H5P_DEFAULT, 0, offset, j.out_bytes[c], j.out[c]) < 0) rc = NCC_EHDFERR;
        }
        free(j.out[c]);
    }
    free(j.out); free(j.out_bytes);
    if (rc == 1) g_chunks_written += j.nchunks;
    return rc;
}

int nch5_put_vara(NcH5 *h, int varid, int nd, const uint64_t *start, const uint64_t *count, int memtype, const void *data) {
    if (!h || varid < 0 || varid >= h->nvars) return NCC_ENOTVAR;
    const hid_t mt = mem_type(memtype);
    if (mt < 0) return NCC_EBADTYPE;
    const hid_t d = h->dset[varid];
    if (nd == 0) return H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0 ? NCC_EHDFERR : 0;
    hid_t fs = H5Dget_space(d);
    if (fs < 0) return NCC_EHDFERR;
    hsize_t st[NCC_MAX_DIMS], ct[NCC_MAX_DIMS], ext[NCC_MAX_DIMS], mx[NCC_MAX_DIMS];
    int rc = 0, grow = 0;
    if (H5Sget_simple_extent_ndims(fs) != nd) rc = NCC_EINVALCOORDS;
    if (!rc) {
        H5Sget_simple_extent_dims(fs, ext, mx);
        for (int i = 0; i < nd; ++i) {
            st[i] = start[i]; ct[i] = count[i];
            if (st[i] + ct[i] > ext[i]) {
                if (mx[i] == H5S_UNLIMITED) { ext[i] = st[i] + ct[i]; grow = 1; }
                else rc = NCC_EEDGE;
            }
        }
    }
    if (!rc && grow) {
        H5Sclose(fs);
        if (H5Dset_extent(d, ext) < 0) return NCC_EHDFERR;
        fs = H5Dget_space(d);
    }
    if (!rc && h->vout) {
        const int w = put_whole_chunks(h, varid, nd, st, ct, ext, memtype, data);
        if (w != 0) { H5Sclose(fs); return w < 0 ? w : 0; }
    }
    if (!rc) {
        hid_t ms = H5Screate_simple(nd, ct, NULL);
        if (ms < 0 || H5Sselect_hyperslab(fs, H5S_SELECT_SET, st, NULL, ct, NULL) < 0 ||
            H5Dwrite(d, mt, ms, fs, H5P_DEFAULT, data) < 0)
            rc = NCC_EHDFERR;
        if (ms >= 0) H5Sclose(ms);
    }
    H5Sclose(fs);
    return rc;
}

int nch5_finish(NcH5 *h, uint64_t numrecs) {
    if (!h) return 0;
    int rc = 0;
    /* every dataset on the record dimension gets its final length (a variable never written reads as fill values) */
    for (int pass = 0; pass < 2; ++pass) {
        const int n = pass ? h->ndims : h->nvars;
        for (int i = 0; i < n; ++i) {
            const hid_t d = pass ? (h->own_scale[i] ? h->scale[i] : -1) : h->dset[i];
            if (d < 0) continue;
            hid_t fs = H5Dget_space(d);
            hsize_t ext[NCC_MAX_DIMS], mx[NCC_MAX_DIMS];
            const int nd = H5Sget_simple_extent_ndims(fs);
            if (nd >= 1) {
                H5Sget_simple_extent_dims(fs, ext, mx);
                if (mx[0] == H5S_UNLIMITED && ext[0] < numrecs) { ext[0] = numrecs; if (H5Dset_extent(d, ext) < 0) rc = NCC_EHDFERR; }
            }
            H5Sclose(fs);
        }
    }
    for (int i = 0; i < h->nvars; ++i) if (h->dset[i] >= 0) H5Dclose(h->dset[i]);
    for (int i = 0; i < h->ndims; ++i) if (h->own_scale[i] && h->scale[i] >= 0) H5Dclose(h->scale[i]);
    if (H5Fclose(h->file) < 0) rc = NCC_EHDFERR;
    free(h->dset); free(h->vout); free(h->scale); free(h->own_scale);
    free(h);
    return rc;
}
