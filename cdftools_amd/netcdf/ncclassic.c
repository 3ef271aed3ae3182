/*
 * ncclassic.c -- NetCDF classic format (CDF-1 / CDF-2 / CDF-5) reader/writer; see ncclassic.h.
 * Written from the published file-format specification (NetCDF User's Guide, "File Format
 * Specification"); no code from libnetcdf.  Single-threaded use, like the reference's cdfio.
 */
#define _FILE_OFFSET_BITS 64
#define _GNU_SOURCE
#include "ncclassic.h"
#include "ncmodel.h"
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#define TAG_DIM 0x0A
#define TAG_VAR 0x0B
#define TAG_ATT 0x0C
#define MAX_FILES 64

typedef struct {
    int used, fd, format, writable, indef, has_data, hdr_dirty, nrec_dirty;
    uint64_t numrecs, recsize, hdr_size; /* hdr_size: bytes reserved for the header in the file */
    int ndims, nvars, ngatts, unlim;
    Dim *dims;
    Var *vars;
    Att *gatts;
    uint64_t *old_begin; /* begins before the current redef (NULL outside redef of a file with data) */
    uint64_t old_recsize;
    NcH5 *h5;            /* a NetCDF-4 file, read or written through the HDF5 plug-in */
    int nc4w;            /* NetCDF-4 output: the HDF5 file is created from the model at the first enddef */
    struct Behind *wb;   /* writes queued by ncc_put_vara_behind and the thread that carries them out (NULL: never used) */
    char path[4096];
} File;

static File g_files[MAX_FILES];

/* Write-behind (ncc_put_vara_behind / ncc_wait_writes): a program that computes slab s+1 while slab s goes to the file queues
 * the writes of slab s and goes on READING; one thread per file carries the queue out in order (each write still split over
 * the writing threads).  Every other entry point on that file first waits for the queue to empty (getf), so nothing else
 * ever sees the file while the thread works on it; the data must stay untouched until ncc_wait_writes (or any other call on
 * the file) returns. */
#include <pthread.h>
typedef struct { int varid, memtype, has_start, has_count; size_t start[NCC_MAX_DIMS + 1], count[NCC_MAX_DIMS + 1]; const void *data; } BehindReq;
enum { kBehindQueue = 64 };
typedef struct Behind {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    BehindReq q[kBehindQueue];
    int head, n, busy, stop, rc;
    File *f;
} Behind;

static void behind_stop(File *f);

static void behind_drain(File *f) {
    Behind *b = f->wb;
    if (!b) return;
    pthread_mutex_lock(&b->mu);
    while (b->n > 0 || b->busy) pthread_cond_wait(&b->cv, &b->mu);
    pthread_mutex_unlock(&b->mu);
}

static File *getf_nowait(int ncid) {
    if (ncid < 0 || ncid >= MAX_FILES || !g_files[ncid].used) return NULL;
    return &g_files[ncid];
}

static File *getf(int ncid) {
    File *f = getf_nowait(ncid);
    if (f && f->wb) behind_drain(f);
    return f;
}

static size_t tsize(int t) {
    switch (t) {
        case NCC_BYTE: case NCC_CHAR: case NCC_UBYTE: return 1;
        case NCC_SHORT: case NCC_USHORT: return 2;
        case NCC_INT: case NCC_UINT: case NCC_FLOAT: return 4;
        case NCC_DOUBLE: case NCC_INT64: case NCC_UINT64: return 8;
        default: return 0;
    }
}

static uint64_t pad4(uint64_t n) { return (n + 3u) & ~(uint64_t)3u; }

/* ------------------------------------------------------------------ byte order + conversion */
/* n elements of w bytes, byte order reversed; dst may be src.  The fixed-width loops vectorise (file built -O3). */
static void bswap_copy(void *dst, const void *src, size_t n, size_t w) {
    if (w == 1) { if (dst != src) memcpy(dst, src, n); return; }
    if (w == 4) {
        const unsigned char *s = (const unsigned char *)src;
        unsigned char *d = (unsigned char *)dst;
        for (size_t i = 0; i < n; ++i) { uint32_t v; memcpy(&v, s + 4 * i, 4); v = __builtin_bswap32(v); memcpy(d + 4 * i, &v, 4); }
        return;
    }
    if (w == 8) {
        const unsigned char *s = (const unsigned char *)src;
        unsigned char *d = (unsigned char *)dst;
        for (size_t i = 0; i < n; ++i) { uint64_t v; memcpy(&v, s + 8 * i, 8); v = __builtin_bswap64(v); memcpy(d + 8 * i, &v, 8); }
        return;
    }
    if (w == 2) {
        const unsigned char *s = (const unsigned char *)src;
        unsigned char *d = (unsigned char *)dst;
        for (size_t i = 0; i < n; ++i) { uint16_t v; memcpy(&v, s + 2 * i, 2); v = __builtin_bswap16(v); memcpy(d + 2 * i, &v, 2); }
        return;
    }
    const unsigned char *s = (const unsigned char *)src;
    unsigned char *d = (unsigned char *)dst;
    for (size_t i = 0; i < n; ++i, s += w, d += w)
        for (size_t b = 0; b < w; ++b) d[b] = s[w - 1 - b];
}

static double load_as_double(const void *p, int t, size_t i) {
    switch (t) {
        case NCC_BYTE: return ((const signed char *)p)[i];
        case NCC_CHAR: case NCC_UBYTE: return ((const unsigned char *)p)[i];
        case NCC_SHORT: return ((const int16_t *)p)[i];
        case NCC_USHORT: return ((const uint16_t *)p)[i];
        case NCC_INT: return ((const int32_t *)p)[i];
        case NCC_UINT: return ((const uint32_t *)p)[i];
        case NCC_FLOAT: return ((const float *)p)[i];
        case NCC_DOUBLE: return ((const double *)p)[i];
        case NCC_INT64: return (double)((const int64_t *)p)[i];
        case NCC_UINT64: return (double)((const uint64_t *)p)[i];
    }
    return 0.0;
}

static void store_from_double(void *p, int t, size_t i, double v) {
    switch (t) {
        case NCC_BYTE: ((signed char *)p)[i] = (signed char)v; break;
        case NCC_CHAR: case NCC_UBYTE: ((unsigned char *)p)[i] = (unsigned char)v; break;
        case NCC_SHORT: ((int16_t *)p)[i] = (int16_t)v; break;
        case NCC_USHORT: ((uint16_t *)p)[i] = (uint16_t)v; break;
        case NCC_INT: ((int32_t *)p)[i] = (int32_t)v; break;
        case NCC_UINT: ((uint32_t *)p)[i] = (uint32_t)v; break;
        case NCC_FLOAT: ((float *)p)[i] = (float)v; break;   /* round to nearest, like the C library */
        case NCC_DOUBLE: ((double *)p)[i] = v; break;
        case NCC_INT64: ((int64_t *)p)[i] = (int64_t)v; break;
        case NCC_UINT64: ((uint64_t *)p)[i] = (uint64_t)v; break;
    }
}

/* native array of type `from` -> native array of type `to` */
static void convert(void *dst, int to, const void *src, int from, size_t n) {
    if (to == from || (tsize(to) == 1 && tsize(from) == 1)) { memcpy(dst, src, n * tsize(to)); return; }
    /* the conversions getvar meets all the time (masks as bytes, r8 or i2 variables read as r4), as plain loops */
    if (to == NCC_FLOAT) {
        float *d = (float *)dst;
        if (from == NCC_BYTE)   { const signed char *q = (const signed char *)src; for (size_t i = 0; i < n; ++i) d[i] = (float)q[i]; return; }
        if (from == NCC_SHORT)  { const int16_t *q = (const int16_t *)src;         for (size_t i = 0; i < n; ++i) d[i] = (float)q[i]; return; }
        if (from == NCC_INT)    { const int32_t *q = (const int32_t *)src;         for (size_t i = 0; i < n; ++i) d[i] = (float)(double)q[i]; return; }
        if (from == NCC_DOUBLE) { const double *q = (const double *)src;           for (size_t i = 0; i < n; ++i) d[i] = (float)q[i]; return; }
    }
    if (to == NCC_DOUBLE && from == NCC_FLOAT) {
        const float *q = (const float *)src; double *d = (double *)dst;
        for (size_t i = 0; i < n; ++i) d[i] = (double)q[i];
        return;
    }
    for (size_t i = 0; i < n; ++i) store_from_double(dst, to, i, load_as_double(src, from, i));
}

/* ------------------------------------------------------------------ growable big-endian buffer */
typedef struct { unsigned char *p; size_t n, cap; } Buf;

static int buf_need(Buf *b, size_t extra) {
    if (b->n + extra <= b->cap) return 0;
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->n + extra) nc *= 2;
    unsigned char *q = (unsigned char *)realloc(b->p, nc);
    if (!q) return NCC_ENOMEM;
    b->p = q; b->cap = nc;
    return 0;
}
static int put_u32(Buf *b, uint32_t v) {
    if (buf_need(b, 4)) return NCC_ENOMEM;
    for (int i = 0; i < 4; ++i) b->p[b->n++] = (unsigned char)(v >> (24 - 8 * i));
    return 0;
}
static int put_u64(Buf *b, uint64_t v) {
    if (buf_need(b, 8)) return NCC_ENOMEM;
    for (int i = 0; i < 8; ++i) b->p[b->n++] = (unsigned char)(v >> (56 - 8 * i));
    return 0;
}
static int put_nn(Buf *b, int fmt, uint64_t v) { return fmt == 5 ? put_u64(b, v) : put_u32(b, (uint32_t)v); }
static int put_name(Buf *b, int fmt, const char *s) {
    size_t n = strlen(s);
    if (put_nn(b, fmt, n) || buf_need(b, pad4(n))) return NCC_ENOMEM;
    memcpy(b->p + b->n, s, n);
    memset(b->p + b->n + n, 0, pad4(n) - n);
    b->n += pad4(n);
    return 0;
}
static int put_atts(Buf *b, int fmt, const Att *a, int na) {
    if (na == 0) return put_u32(b, 0) || put_nn(b, fmt, 0);
    if (put_u32(b, TAG_ATT) || put_nn(b, fmt, na)) return NCC_ENOMEM;
    for (int i = 0; i < na; ++i) {
        const size_t w = tsize(a[i].type), nb = a[i].nelems * w;
        if (put_name(b, fmt, a[i].name) || put_u32(b, a[i].type) || put_nn(b, fmt, a[i].nelems) || buf_need(b, pad4(nb)))
            return NCC_ENOMEM;
        bswap_copy(b->p + b->n, a[i].data, a[i].nelems, w);
        memset(b->p + b->n + nb, 0, pad4(nb) - nb);
        b->n += pad4(nb);
    }
    return 0;
}

static int serialize_header(const File *f, Buf *b) {
    const int fmt = f->format;
    if (buf_need(b, 4)) return NCC_ENOMEM;
    b->p[b->n++] = 'C'; b->p[b->n++] = 'D'; b->p[b->n++] = 'F'; b->p[b->n++] = (unsigned char)fmt;
    if (put_nn(b, fmt, f->numrecs)) return NCC_ENOMEM;
    if (f->ndims == 0) { if (put_u32(b, 0) || put_nn(b, fmt, 0)) return NCC_ENOMEM; }
    else {
        if (put_u32(b, TAG_DIM) || put_nn(b, fmt, f->ndims)) return NCC_ENOMEM;
        for (int i = 0; i < f->ndims; ++i)
            if (put_name(b, fmt, f->dims[i].name) || put_nn(b, fmt, f->dims[i].len)) return NCC_ENOMEM;
    }
    if (put_atts(b, fmt, f->gatts, f->ngatts)) return NCC_ENOMEM;
    if (f->nvars == 0) { if (put_u32(b, 0) || put_nn(b, fmt, 0)) return NCC_ENOMEM; }
    else {
        if (put_u32(b, TAG_VAR) || put_nn(b, fmt, f->nvars)) return NCC_ENOMEM;
        for (int i = 0; i < f->nvars; ++i) {
            const Var *v = &f->vars[i];
            if (put_name(b, fmt, v->name) || put_nn(b, fmt, v->ndims)) return NCC_ENOMEM;
            for (int d = 0; d < v->ndims; ++d)
                if (put_nn(b, fmt, v->dimids[d])) return NCC_ENOMEM;
            if (put_atts(b, fmt, v->atts, v->natts) || put_u32(b, v->type)) return NCC_ENOMEM;
            uint64_t vs = v->vsize;
            if (fmt != 5 && vs > 0xFFFFFFFFull) vs = 0xFFFFFFFFull;
            if (put_nn(b, fmt, vs)) return NCC_ENOMEM;
            if (fmt == 1) { if (put_u32(b, (uint32_t)v->begin)) return NCC_ENOMEM; }
            else if (put_u64(b, v->begin)) return NCC_ENOMEM;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ header parsing */
typedef struct { const unsigned char *p; size_t n, pos; int fmt; int err; } Rd;
static uint32_t rd_u32(Rd *r) {
    if (r->pos + 4 > r->n) { r->err = 1; return 0; }
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) v = (v << 8) | r->p[r->pos++];
    return v;
}
static uint64_t rd_u64(Rd *r) {
    if (r->pos + 8 > r->n) { r->err = 1; return 0; }
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v = (v << 8) | r->p[r->pos++];
    return v;
}
static uint64_t rd_nn(Rd *r) { return r->fmt == 5 ? rd_u64(r) : rd_u32(r); }
static void rd_name(Rd *r, char *out) {
    uint64_t n = rd_nn(r);
    if (r->err || n >= NCC_MAX_NAME || r->pos + pad4(n) > r->n) { r->err = 1; out[0] = 0; return; }
    memcpy(out, r->p + r->pos, n);
    out[n] = 0;
    r->pos += pad4(n);
}
static int rd_atts(Rd *r, Att **out, int *nout) {
    uint32_t tag = rd_u32(r);
    uint64_t n = rd_nn(r);
    *out = NULL; *nout = 0;
    if (r->err) return NCC_ENOTNC;
    if (tag == 0 && n == 0) return 0;
    if (tag != TAG_ATT) { r->err = 1; return NCC_ENOTNC; }
    Att *a = (Att *)calloc(n ? n : 1, sizeof(Att));
    if (!a) return NCC_ENOMEM;
    for (uint64_t i = 0; i < n; ++i) {
        rd_name(r, a[i].name);
        a[i].type = (int)rd_u32(r);
        a[i].nelems = rd_nn(r);
        const size_t w = tsize(a[i].type), nb = a[i].nelems * w;
        if (r->err || w == 0 || r->pos + pad4(nb) > r->n) { r->err = 1; break; }
        a[i].data = malloc(nb ? nb : 1);
        bswap_copy(a[i].data, r->p + r->pos, a[i].nelems, w);
        r->pos += pad4(nb);
    }
    *out = a; *nout = (int)n;
    return r->err ? NCC_ENOTNC : 0;
}

static void compute_sizes(File *f) {
    /* vsize of every variable, record flags, recsize (with the one-record-variable exception) */
    int nrecvars = 0;
    for (int i = 0; i < f->nvars; ++i) {
        Var *v = &f->vars[i];
        uint64_t n = tsize(v->type);
        v->isrec = (v->ndims > 0 && f->unlim >= 0 && v->dimids[0] == f->unlim);
        for (int d = v->isrec ? 1 : 0; d < v->ndims; ++d) n *= f->dims[v->dimids[d]].len;
        v->vsize = pad4(n);
        if (v->isrec) ++nrecvars;
    }
    f->recsize = 0;
    for (int i = 0; i < f->nvars; ++i) {
        Var *v = &f->vars[i];
        if (!v->isrec) continue;
        if (nrecvars == 1) {             /* a single record variable is not padded */
            uint64_t n = tsize(v->type);
            for (int d = 1; d < v->ndims; ++d) n *= f->dims[v->dimids[d]].len;
            f->recsize = n;
        } else {
            f->recsize += v->vsize;
        }
    }
}

static int parse_header(File *f) {
    /* read progressively larger prefixes until the header parses */
    size_t cap = 1 << 16;
    for (;;) {
        unsigned char *buf = (unsigned char *)malloc(cap);
        if (!buf) return NCC_ENOMEM;
        ssize_t got = pread(f->fd, buf, cap, 0);
        if (got < 4) { free(buf); return NCC_ENOTNC; }
        if (buf[0] != 'C' || buf[1] != 'D' || buf[2] != 'F' || (buf[3] != 1 && buf[3] != 2 && buf[3] != 5)) {
            free(buf);
            return NCC_ENOTNC;
        }
        Rd r = {buf, (size_t)got, 4, buf[3], 0};
        f->format = buf[3];
        f->numrecs = rd_nn(&r);
        if (f->format != 5 && f->numrecs == 0xFFFFFFFFull) f->numrecs = 0;   /* STREAMING: recomputed below */
        uint32_t tag = rd_u32(&r);
        uint64_t n = rd_nn(&r);
        f->unlim = -1;
        f->ndims = 0; f->dims = NULL;
        if (!(tag == 0 && n == 0)) {
            if (tag != TAG_DIM) r.err = 1;
            else {
                f->dims = (Dim *)calloc(n ? n : 1, sizeof(Dim));
                f->ndims = (int)n;
                for (uint64_t i = 0; i < n && !r.err; ++i) {
                    rd_name(&r, f->dims[i].name);
                    f->dims[i].len = rd_nn(&r);
                    if (f->dims[i].len == 0) f->unlim = (int)i;
                }
            }
        }
        int rc = r.err ? NCC_ENOTNC : rd_atts(&r, &f->gatts, &f->ngatts);
        f->nvars = 0; f->vars = NULL;
        if (!rc) {
            tag = rd_u32(&r);
            n = rd_nn(&r);
            if (!(tag == 0 && n == 0)) {
                if (tag != TAG_VAR) r.err = 1;
                else {
                    f->vars = (Var *)calloc(n ? n : 1, sizeof(Var));
                    f->nvars = (int)n;
                    for (uint64_t i = 0; i < n && !r.err; ++i) {
                        Var *v = &f->vars[i];
                        rd_name(&r, v->name);
                        v->ndims = (int)rd_nn(&r);
                        if (v->ndims > NCC_MAX_DIMS) { r.err = 1; break; }
                        for (int d = 0; d < v->ndims; ++d) v->dimids[d] = (int)rd_nn(&r);
                        if (rd_atts(&r, &v->atts, &v->natts)) { r.err = 1; break; }
                        v->type = (int)rd_u32(&r);
                        (void)rd_nn(&r); /* vsize: recomputed */
                        v->begin = (f->format == 1) ? rd_u32(&r) : rd_u64(&r);
                    }
                }
            }
        }
        const int truncated = r.err && (size_t)got == cap;
        const size_t hdr = r.pos;
        free(buf);
        if (truncated) {   /* header longer than the prefix: free what was built and retry */
            for (int i = 0; i < f->nvars; ++i) { for (int a = 0; a < f->vars[i].natts; ++a) free(f->vars[i].atts[a].data); free(f->vars[i].atts); }
            for (int a = 0; a < f->ngatts; ++a) free(f->gatts[a].data);
            free(f->vars); free(f->gatts); free(f->dims);
            f->vars = NULL; f->gatts = NULL; f->dims = NULL;
            cap *= 8;
            if (cap > ((size_t)1 << 30)) return NCC_ENOTNC;
            continue;
        }
        if (r.err || rc) return NCC_ENOTNC;
        compute_sizes(f);
        uint64_t first = UINT64_MAX;
        for (int i = 0; i < f->nvars; ++i) if (f->vars[i].begin < first) first = f->vars[i].begin;
        f->hdr_size = (f->nvars && first != UINT64_MAX) ? first : hdr;
        return 0;
    }
}

/* ------------------------------------------------------------------ layout */
static int write_header(File *f) {
    Buf b = {0, 0, 0};
    int rc = serialize_header(f, &b);
    if (!rc && b.n > f->hdr_size && f->nvars) rc = NCC_EINVAL;   /* cannot happen after layout() */
    if (!rc && pwrite(f->fd, b.p, b.n, 0) != (ssize_t)b.n) rc = NCC_EIO;
    free(b.p);
    f->hdr_dirty = 0; f->nrec_dirty = 0;
    return rc;
}

static uint64_t header_bytes(const File *f) {
    Buf b = {0, 0, 0};
    uint64_t n = serialize_header(f, &b) ? 0 : b.n;
    free(b.p);
    return n;
}

static int move_block(int fd, uint64_t from, uint64_t to, uint64_t n) {
    if (from == to || n == 0) return 0;
    const size_t chunk = 8u << 20;
    unsigned char *tmp = (unsigned char *)malloc(chunk);
    if (!tmp) return NCC_ENOMEM;
    int rc = 0;
    if (to > from) {                       /* move towards the end: last chunk first */
        uint64_t left = n;
        while (left && !rc) {
            const size_t c = left < chunk ? (size_t)left : chunk;
            left -= c;
            ssize_t g = pread(fd, tmp, c, (off_t)(from + left));
            if (g < 0) rc = NCC_EIO;
            else { if ((size_t)g < c) memset(tmp + g, 0, c - g); if (pwrite(fd, tmp, c, (off_t)(to + left)) != (ssize_t)c) rc = NCC_EIO; }
        }
    } else {
        uint64_t done = 0;
        while (done < n && !rc) {
            const size_t c = n - done < chunk ? (size_t)(n - done) : chunk;
            ssize_t g = pread(fd, tmp, c, (off_t)(from + done));
            if (g < 0) rc = NCC_EIO;
            else { if ((size_t)g < c) memset(tmp + g, 0, c - g); if (pwrite(fd, tmp, c, (off_t)(to + done)) != (ssize_t)c) rc = NCC_EIO; }
            done += c;
        }
    }
    free(tmp);
    return rc;
}

/* assign begins; when the file already holds data, move it to the new places */
static int layout(File *f) {
    compute_sizes(f);
    const uint64_t need = header_bytes(f);
    if (!need) return NCC_ENOMEM;
    uint64_t start = f->hdr_size;
    if (!f->has_data || need > start) start = pad4(need) + (f->has_data ? 4096 : 0);   /* slack once we have to move */
    if (f->has_data && !f->old_begin) return NCC_EINVAL;
    uint64_t off = start;
    uint64_t *nb = (uint64_t *)calloc(f->nvars ? f->nvars : 1, sizeof(uint64_t));
    for (int i = 0; i < f->nvars; ++i) if (!f->vars[i].isrec) { nb[i] = off; off += f->vars[i].vsize; }
    const uint64_t rec0 = off;
    for (int i = 0; i < f->nvars; ++i) if (f->vars[i].isrec) { nb[i] = off; off += f->vars[i].vsize; }
    int rc = 0;
    /* The classic formats' size rules (NetCDF User's Guide, "Large File Support"): the header stores vsize in 32 bits
     * (CDF-1, CDF-2) and begin in 32 bits (CDF-1).  A fixed-size variable, or one record of a record variable, must
     * fit 2^31 - 4 (CDF-1) / 2^32 - 4 (CDF-2) bytes unless it is the LAST fixed-size variable of a file without record
     * variables, or the last record variable; every begin of a CDF-1 file must fit 31 bits.  Anything else would be
     * written with a clamped vsize or a truncated offset that other readers misinterpret: refuse it here (cdfio's
     * create() switches to CDF-5 before that happens). */
    if (f->format != 5) {
        const uint64_t vmax = f->format == 1 ? 0x7FFFFFFCull : 0xFFFFFFFCull;
        int last_fix = -1, last_rec = -1;
        for (int i = 0; i < f->nvars; ++i) { if (f->vars[i].isrec) last_rec = i; else last_fix = i; }
        for (int i = 0; i < f->nvars && !rc; ++i) {
            const int may_be_big = f->vars[i].isrec ? (i == last_rec) : (i == last_fix && last_rec < 0);
            if (f->vars[i].vsize > vmax && !may_be_big) rc = NCC_EVARSIZE;
            if (f->format == 1 && nb[i] > 0x7FFFFFFFull) rc = NCC_EVARSIZE;
        }
        if (rc) { free(nb); return rc; }
    }
    if (f->has_data) {
        /* only layouts that keep the record structure can be moved in place */
        if (f->numrecs > 0 && f->recsize != f->old_recsize) rc = NCC_EINVAL;
        /* record section first (it sits last), then fixed variables from the last to the first */
        uint64_t old_rec0 = UINT64_MAX;
        for (int i = 0; i < f->nvars; ++i) if (f->vars[i].isrec && f->old_begin[i] != UINT64_MAX && f->old_begin[i] < old_rec0) old_rec0 = f->old_begin[i];
        if (!rc && old_rec0 != UINT64_MAX && f->numrecs > 0) rc = move_block(f->fd, old_rec0, rec0, f->numrecs * f->recsize);
        for (int i = f->nvars - 1; i >= 0 && !rc; --i)
            if (!f->vars[i].isrec && f->old_begin[i] != UINT64_MAX) rc = move_block(f->fd, f->old_begin[i], nb[i], f->vars[i].vsize);
    }
    for (int i = 0; i < f->nvars; ++i) f->vars[i].begin = nb[i];
    free(nb);
    f->hdr_size = start;
    return rc;
}

/* ------------------------------------------------------------------ open / create / close */
static int new_slot(void) {
    for (int i = 0; i < MAX_FILES; ++i) if (!g_files[i].used) return i;
    return -1;
}

static int h5_load(void);

int ncc_create(const char *path, int format, int clobber, int *ncid) {
    if (!path || !ncid || (format != 1 && format != 2 && format != 3 && format != 5)) return NCC_EINVAL;
    const int s = new_slot();
    if (s < 0) return NCC_ENOMEM;
    if (format == NCC_FORMAT_NETCDF4) {                /* the HDF5 file itself is created at the first enddef */
        if (strlen(path) >= sizeof g_files[s].path) return NCC_EINVAL;
        const int rc4 = h5_load();
        if (rc4) return rc4;
        if (!clobber && access(path, F_OK) == 0) return NCC_EEXIST;
        File *f4 = &g_files[s];
        memset(f4, 0, sizeof *f4);
        f4->used = 1; f4->fd = -1; f4->format = format; f4->writable = 1; f4->indef = 1; f4->unlim = -1; f4->nc4w = 1;
        strcpy(f4->path, path);
        *ncid = s;
        return 0;
    }
    const int fd = open(path, O_RDWR | O_CREAT | (clobber ? O_TRUNC : O_EXCL), 0644);
    if (fd < 0) return errno == EEXIST ? NCC_EEXIST : NCC_ENOENT;
    File *f = &g_files[s];
    memset(f, 0, sizeof *f);
    f->used = 1; f->fd = fd; f->format = format; f->writable = 1; f->indef = 1; f->unlim = -1; f->hdr_dirty = 1;
    *ncid = s;
    return 0;
}

/* The HDF5 plug-in lives next to this library (libnch5.so; CDFT_NCH5_LIB overrides) and is only loaded when a
 * NetCDF-4 file is opened, so that the classic path does not depend on an HDF5 installation. */
static nch5_open_fn g_h5_open;
static nch5_get_vara_fn g_h5_get;
static nch5_close_fn g_h5_close;
static nch5_create_fn g_h5_create;
static nch5_put_vara_fn g_h5_put;
static nch5_put_atts_fn g_h5_put_atts;
static nch5_finish_fn g_h5_finish;
static char g_h5_msg[512] = "NetCDF: HDF5 error";

static int h5_load(void) {
    if (g_h5_open) return 0;
    char path[4096] = "libnch5.so";
    const char *env = getenv("CDFT_NCH5_LIB");
    Dl_info info;
    if (env && *env) snprintf(path, sizeof path, "%s", env);
    else if (dladdr((void *)&h5_load, &info) && info.dli_fname) {
        snprintf(path, sizeof path, "%s", info.dli_fname);
        char *slash = strrchr(path, '/');
        if (slash) snprintf(slash + 1, sizeof path - (size_t)(slash + 1 - path), "libnch5.so");
        else snprintf(path, sizeof path, "libnch5.so");
    }
    void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        snprintf(g_h5_msg, sizeof g_h5_msg, "NetCDF: a NetCDF-4/HDF5 file needs the HDF5 plug-in, which could not be loaded (%s)",
                 dlerror());
        return NCC_EHDFERR;
    }
    g_h5_open  = (nch5_open_fn)dlsym(lib, "nch5_open");
    g_h5_get   = (nch5_get_vara_fn)dlsym(lib, "nch5_get_vara");
    g_h5_close = (nch5_close_fn)dlsym(lib, "nch5_close");
    g_h5_create = (nch5_create_fn)dlsym(lib, "nch5_create");
    g_h5_put = (nch5_put_vara_fn)dlsym(lib, "nch5_put_vara");
    g_h5_put_atts = (nch5_put_atts_fn)dlsym(lib, "nch5_put_atts");
    g_h5_finish = (nch5_finish_fn)dlsym(lib, "nch5_finish");
    if (!g_h5_open || !g_h5_get || !g_h5_close || !g_h5_create || !g_h5_put || !g_h5_put_atts || !g_h5_finish) {
        g_h5_open = NULL;
        snprintf(g_h5_msg, sizeof g_h5_msg, "NetCDF: %.400s is not the HDF5 plug-in of this build", path);
        return NCC_EHDFERR;
    }
    return 0;
}

int ncc_open(const char *path, int writable, int *ncid) {
    if (!path || !ncid) return NCC_EINVAL;
    const int s = new_slot();
    if (s < 0) return NCC_ENOMEM;
    const int fd = open(path, writable ? O_RDWR : O_RDONLY);
    if (fd < 0) return NCC_ENOENT;
    File *f = &g_files[s];
    memset(f, 0, sizeof *f);
    f->used = 1; f->fd = fd; f->writable = writable; f->has_data = 1;
    unsigned char sig[8] = {0};
    if (pread(fd, sig, 8, 0) == 8 && memcmp(sig, "\211HDF\r\n\032\n", 8) == 0) {      /* NetCDF-4 = HDF5 */
        close(fd);
        f->fd = -1;
        int rc4 = writable ? NCC_EPERM : h5_load();
        if (!rc4) rc4 = g_h5_open(path, &f->h5, &f->ndims, &f->dims, &f->nvars, &f->vars, &f->ngatts, &f->gatts, &f->unlim,
                                  &f->numrecs);
        if (rc4) { f->used = 0; return rc4; }
        f->format = NCC_FORMAT_NETCDF4;
        *ncid = s;
        return 0;
    }
    const int rc = parse_header(f);
    if (rc) { close(fd); f->used = 0; return rc; }
    if (f->recsize && f->numrecs == 0) {           /* numrecs not flushed (or STREAMING): derive from the size */
        struct stat st;
        uint64_t rec0 = UINT64_MAX;
        for (int i = 0; i < f->nvars; ++i) if (f->vars[i].isrec && f->vars[i].begin < rec0) rec0 = f->vars[i].begin;
        if (fstat(fd, &st) == 0 && rec0 != UINT64_MAX && (uint64_t)st.st_size > rec0)
            f->numrecs = ((uint64_t)st.st_size - rec0) / f->recsize;
    }
    *ncid = s;
    return 0;
}

static void free_file(File *f) {
    behind_stop(f);
    for (int i = 0; i < f->nvars; ++i) { for (int a = 0; a < f->vars[i].natts; ++a) free(f->vars[i].atts[a].data); free(f->vars[i].atts); }
    for (int a = 0; a < f->ngatts; ++a) free(f->gatts[a].data);
    free(f->vars); free(f->gatts); free(f->dims); free(f->old_begin);
    memset(f, 0, sizeof *f);
}

int ncc_enddef(int ncid) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (!f->indef) return NCC_ENOTINDEFINE;
    if (f->nc4w) {
        int rc4 = 0;
        if (!f->h5) {
            for (int i = 0; i < f->nvars; ++i) f->vars[i].isrec = (f->vars[i].ndims > 0 && f->vars[i].dimids[0] == f->unlim);
            rc4 = g_h5_create(f->path, &f->h5, f->ndims, f->dims, f->nvars, f->vars, f->ngatts, f->gatts, f->unlim);
        } else {                                       /* attributes added or changed in this define mode */
            rc4 = g_h5_put_atts(f->h5, NCC_GLOBAL, f->ngatts, f->gatts);
            for (int i = 0; i < f->nvars && !rc4; ++i) rc4 = g_h5_put_atts(f->h5, i, f->vars[i].natts, f->vars[i].atts);
        }
        f->indef = 0;
        return rc4;
    }
    int rc = layout(f);
    if (!rc) rc = write_header(f);
    free(f->old_begin);
    f->old_begin = NULL;
    f->indef = 0;
    return rc;
}

int ncc_redef(int ncid) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (!f->writable) return NCC_EPERM;
    if (f->indef) return NCC_EINDEFINE;
    f->indef = 1;
    if (f->nc4w) return 0;
    free(f->old_begin);
    f->old_begin = (uint64_t *)calloc(f->nvars + 1024, sizeof(uint64_t));
    for (int i = 0; i < f->nvars + 1024; ++i) f->old_begin[i] = i < f->nvars ? f->vars[i].begin : UINT64_MAX;
    f->old_recsize = f->recsize;
    return 0;
}

int ncc_close(int ncid) {
    File *f = getf(ncid);                               /* (waits for the queued writes) */
    if (!f) return NCC_EBADID;
    /* a queued write that failed and that no ncc_wait_writes has reported yet: the close reports it, as the synchronous
     * ncc_put_vara would have */
    int rc = f->wb ? f->wb->rc : 0;
    if (f->nc4w) {
        int rc4 = f->indef ? ncc_enddef(ncid) : 0;     /* a file that never left define mode is still created */
        if (f->h5) { const int r2 = g_h5_finish(f->h5, f->numrecs); if (!rc4) rc4 = r2; }
        free_file(f);
        return rc4;
    }
    if (f->h5) {
        g_h5_close(f->h5);
        free_file(f);
        return 0;
    }
    if (f->writable) {
        int rh = 0;
        if (f->indef) rh = ncc_enddef(ncid);
        else if (f->hdr_dirty || f->nrec_dirty) rh = write_header(f);
        if (!rc) rc = rh;
        /* the file must extend over every fixed variable and every complete record */
        uint64_t end = f->hdr_size;
        for (int i = 0; i < f->nvars; ++i) {
            const Var *v = &f->vars[i];
            if (!v->isrec && v->begin + v->vsize > end) end = v->begin + v->vsize;
        }
        uint64_t rec0 = UINT64_MAX;
        for (int i = 0; i < f->nvars; ++i) if (f->vars[i].isrec && f->vars[i].begin < rec0) rec0 = f->vars[i].begin;
        if (rec0 != UINT64_MAX && rec0 + f->numrecs * f->recsize > end) end = rec0 + f->numrecs * f->recsize;
        struct stat st;
        if (fstat(f->fd, &st) == 0 && (uint64_t)st.st_size < end && ftruncate(f->fd, (off_t)end) != 0 && !rc) rc = NCC_EIO;
    }
    close(f->fd);
    free_file(f);
    return rc;
}

int ncc_inq_format(int ncid, int *format) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (format) *format = f->format;
    return 0;
}

/* ------------------------------------------------------------------ define */
int ncc_def_dim(int ncid, const char *name, size_t len, int *dimid) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (!f->indef) return NCC_ENOTINDEFINE;
    if (f->nc4w && f->h5) return NCC_EPERM;            /* NetCDF-4 output: everything is defined before the first enddef */
    if (!name || strlen(name) >= NCC_MAX_NAME) return NCC_EINVAL;
    for (int i = 0; i < f->ndims; ++i) if (!strcmp(f->dims[i].name, name)) return NCC_ENAMEINUSE;
    if (len == 0 && f->unlim >= 0) return NCC_EUNLIMIT;
    Dim *d = (Dim *)realloc(f->dims, (f->ndims + 1) * sizeof(Dim));
    if (!d) return NCC_ENOMEM;
    f->dims = d;
    memset(&d[f->ndims], 0, sizeof(Dim));
    strcpy(d[f->ndims].name, name);
    d[f->ndims].len = len;
    if (len == 0) f->unlim = f->ndims;
    if (dimid) *dimid = f->ndims;
    f->ndims++;
    f->hdr_dirty = 1;
    return 0;
}

int ncc_def_var(int ncid, const char *name, int xtype, int ndims, const int *dimids, int *varid) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (!f->indef) return NCC_ENOTINDEFINE;
    if (f->nc4w && f->h5) return NCC_EPERM;
    if (!name || strlen(name) >= NCC_MAX_NAME || ndims < 0 || ndims > NCC_MAX_DIMS) return NCC_EINVAL;
    if (tsize(xtype) == 0 || (xtype > NCC_DOUBLE && f->format != 5 && !f->nc4w)) return NCC_EBADTYPE;
    if (f->nc4w && xtype == NCC_CHAR) return NCC_EBADTYPE;       /* text variables are not written to NetCDF-4 */
    for (int i = 0; i < f->nvars; ++i) if (!strcmp(f->vars[i].name, name)) return NCC_ENAMEINUSE;
    for (int d = 0; d < ndims; ++d) {
        if (dimids[d] < 0 || dimids[d] >= f->ndims) return NCC_EBADDIM;
        if (d > 0 && dimids[d] == f->unlim) return NCC_EUNLIMIT;   /* the record dimension must be the slowest */
    }
    if (f->has_data && f->numrecs > 0 && ndims > 0 && dimids[0] == f->unlim) return NCC_EINVAL;
    if (f->has_data && f->old_begin && f->nvars >= 1024) return NCC_ENOMEM;
    Var *v = (Var *)realloc(f->vars, (f->nvars + 1) * sizeof(Var));
    if (!v) return NCC_ENOMEM;
    f->vars = v;
    memset(&v[f->nvars], 0, sizeof(Var));
    strcpy(v[f->nvars].name, name);
    v[f->nvars].type = xtype;
    v[f->nvars].ndims = ndims;
    for (int d = 0; d < ndims; ++d) v[f->nvars].dimids[d] = dimids[d];
    if (varid) *varid = f->nvars;
    f->nvars++;
    f->hdr_dirty = 1;
    return 0;
}

/* NetCDF-4 output only (ignored for the classic formats, like a libnetcdf call on a classic file would fail) */
int ncc_def_var_chunking(int ncid, int varid, const size_t *chunks) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (!f->indef) return NCC_ENOTINDEFINE;
    if (varid < 0 || varid >= f->nvars) return NCC_ENOTVAR;
    if (!f->nc4w) return 0;
    for (int d = 0; d < f->vars[varid].ndims; ++d) f->vars[varid].chunk[d] = chunks ? chunks[d] : 0;
    return 0;
}

int ncc_def_var_deflate(int ncid, int varid, int shuffle, int level) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (!f->indef) return NCC_ENOTINDEFINE;
    if (varid < 0 || varid >= f->nvars) return NCC_ENOTVAR;
    if (level < 0 || level > 9) return NCC_EINVAL;
    if (!f->nc4w) return 0;
    f->vars[varid].deflate = level;
    f->vars[varid].shuffle = shuffle != 0;
    return 0;
}

/* ------------------------------------------------------------------ inquire */
int ncc_inq(int ncid, int *ndims, int *nvars, int *ngatts, int *unlimdimid) {
    File *f = getf_nowait(ncid);                       /* (as ncc_inq_var) */
    if (!f) return NCC_EBADID;
    if (ndims) *ndims = f->ndims;
    if (nvars) *nvars = f->nvars;
    if (ngatts) *ngatts = f->ngatts;
    if (unlimdimid) *unlimdimid = f->unlim;
    return 0;
}

int ncc_inq_dimid(int ncid, const char *name, int *dimid) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    for (int i = 0; i < f->ndims; ++i)
        if (!strcmp(f->dims[i].name, name)) { if (dimid) *dimid = i; return 0; }
    return NCC_EBADDIM;
}

int ncc_inq_dim(int ncid, int dimid, char *name, size_t *len) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (dimid < 0 || dimid >= f->ndims) return NCC_EBADDIM;
    if (name) strcpy(name, f->dims[dimid].name);
    if (len) *len = (dimid == f->unlim) ? (size_t)f->numrecs : (size_t)f->dims[dimid].len;
    return 0;
}

int ncc_inq_varid(int ncid, const char *name, int *varid) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    for (int i = 0; i < f->nvars; ++i)
        if (!strcmp(f->vars[i].name, name)) { if (varid) *varid = i; return 0; }
    return NCC_ENOTVAR;
}

int ncc_inq_var(int ncid, int varid, char *name, int *xtype, int *ndims, int *dimids, int *natts) {
    File *f = getf_nowait(ncid);                       /* (what the queued writes never change: asked for between two of them) */
    if (!f) return NCC_EBADID;
    if (varid < 0 || varid >= f->nvars) return NCC_ENOTVAR;
    const Var *v = &f->vars[varid];
    if (name) strcpy(name, v->name);
    if (xtype) *xtype = v->type;
    if (ndims) *ndims = v->ndims;
    if (dimids) for (int d = 0; d < v->ndims; ++d) dimids[d] = v->dimids[d];
    if (natts) *natts = v->natts;
    return 0;
}

/* ------------------------------------------------------------------ attributes */
static int att_list(File *f, int varid, Att ***list, int **n) {
    if (varid == NCC_GLOBAL) { *list = &f->gatts; *n = &f->ngatts; return 0; }
    if (varid < 0 || varid >= f->nvars) return NCC_ENOTVAR;
    *list = &f->vars[varid].atts; *n = &f->vars[varid].natts;
    return 0;
}

static Att *find_att(File *f, int varid, const char *name) {
    Att **l; int *n;
    if (att_list(f, varid, &l, &n)) return NULL;
    for (int i = 0; i < *n; ++i) if (!strcmp((*l)[i].name, name)) return &(*l)[i];
    return NULL;
}

int ncc_inq_att(int ncid, int varid, const char *name, int *xtype, size_t *len) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (varid != NCC_GLOBAL && (varid < 0 || varid >= f->nvars)) return NCC_ENOTVAR;
    const Att *a = find_att(f, varid, name);
    if (!a) return NCC_ENOTATT;
    if (xtype) *xtype = a->type;
    if (len) *len = a->nelems;
    return 0;
}

int ncc_inq_attname(int ncid, int varid, int attnum, char *name) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    Att **l; int *n;
    const int rc = att_list(f, varid, &l, &n);
    if (rc) return rc;
    if (attnum < 0 || attnum >= *n) return NCC_ENOTATT;
    strcpy(name, (*l)[attnum].name);
    return 0;
}

int ncc_put_att(int ncid, int varid, const char *name, int xtype, size_t len, int memtype, const void *data) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (!f->writable) return NCC_EPERM;
    if (!name || strlen(name) >= NCC_MAX_NAME || tsize(xtype) == 0 || tsize(memtype) == 0) return NCC_EINVAL;
    if ((xtype == NCC_CHAR) != (memtype == NCC_CHAR)) return NCC_ECHAR;
    Att **l; int *n;
    const int rc = att_list(f, varid, &l, &n);
    if (rc) return rc;
    Att *a = find_att(f, varid, name);
    if (!f->indef && !f->nc4w) {   /* outside define mode an attribute may only be overwritten in place */
        if (!a || pad4(a->nelems * tsize(a->type)) < pad4(len * tsize(xtype))) return NCC_ENOTINDEFINE;
    }
    if (!a) {
        Att *q = (Att *)realloc(*l, (*n + 1) * sizeof(Att));
        if (!q) return NCC_ENOMEM;
        *l = q;
        a = &q[*n];
        memset(a, 0, sizeof *a);
        strcpy(a->name, name);
        (*n)++;
    }
    void *d = malloc(len * tsize(xtype) + 8);
    if (!d) return NCC_ENOMEM;
    convert(d, xtype, data, memtype, len);
    free(a->data);
    a->data = d; a->type = xtype; a->nelems = len;
    f->hdr_dirty = 1;
    if (f->nc4w) {                                     /* after the file exists an attribute goes straight to it */
        if (!f->h5 || f->indef) return 0;
        Att **l2; int *n2;
        att_list(f, varid, &l2, &n2);
        return g_h5_put_atts(f->h5, varid, *n2, *l2);
    }
    if (!f->indef) return write_header(f);
    return 0;
}

int ncc_get_att(int ncid, int varid, const char *name, int memtype, void *data) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (varid != NCC_GLOBAL && (varid < 0 || varid >= f->nvars)) return NCC_ENOTVAR;
    const Att *a = find_att(f, varid, name);
    if (!a) return NCC_ENOTATT;
    if ((a->type == NCC_CHAR) != (memtype == NCC_CHAR)) return NCC_ECHAR;
    convert(data, memtype, a->data, a->type, a->nelems);
    return 0;
}

int ncc_rename_att(int ncid, int varid, const char *name, const char *newname) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    if (!f->writable) return NCC_EPERM;
    Att *a = find_att(f, varid, name);
    if (!a) return NCC_ENOTATT;
    if (strlen(newname) >= NCC_MAX_NAME || find_att(f, varid, newname)) return NCC_ENAMEINUSE;
    if (!f->indef && pad4(strlen(newname)) > pad4(strlen(name))) return NCC_ENOTINDEFINE;
    strcpy(a->name, newname);
    f->hdr_dirty = 1;
    if (!f->indef) return write_header(f);
    return 0;
}

int ncc_copy_att(int ncid_in, int varid_in, const char *name, int ncid_out, int varid_out) {
    File *fi = getf(ncid_in);
    if (!fi) return NCC_EBADID;
    const Att *a = find_att(fi, varid_in, name);
    if (!a) return NCC_ENOTATT;
    return ncc_put_att(ncid_out, varid_out, name, a->type, a->nelems, a->type, a->data);
}

/* ------------------------------------------------------------------ hyperslab I/O */
/* pread until n bytes or end of file (a single pread stops at 2 GiB - 4 KiB on Linux) */
static ssize_t pread_all(int fd, void *buf, size_t n, off_t off) {
    size_t got = 0;
    while (got < n) {
        const ssize_t g = pread(fd, (char *)buf + got, n - got, off + (off_t)got);
        if (g < 0) { if (errno == EINTR) continue; return -1; }
        if (g == 0) break;
        got += (size_t)g;
    }
    return (ssize_t)got;
}

/* A large same-type read is split over a few threads, each reading and byte-swapping its own part: one thread
 * moves ~2 GB/s out of the page cache, and the tools read whole records (hundreds of MB) at a time. */
#include <pthread.h>
/* (xtype != memtype: a conversion on the way -- a byte mask read as r4, an r8 variable as r4 -- done from the same cache-resident
 * buffer, so a mask read as r4 costs its bytes read and its floats written, on every thread; ms = size of the memory type) */
typedef struct { int fd; unsigned char *mem; size_t bytes, xs; off_t off; ssize_t got; int xtype, memtype; size_t ms; } ReadJob;

static void *read_job(void *arg) {
    ReadJob *j = (ReadJob *)arg;
    if (j->xtype != j->memtype) {                        /* read, swap in the buffer, convert into place */
        enum { kCBuf = 256 * 1024 };
        unsigned char *cb = (unsigned char *)malloc(kCBuf);
        j->got = 0;
        if (!cb) { j->got = -1; return NULL; }
        const size_t step = (size_t)kCBuf / j->xs * j->xs;
        for (size_t done = 0; done < j->bytes; done += step) {
            const size_t n = j->bytes - done < step ? j->bytes - done : step;
            const ssize_t g = pread_all(j->fd, cb, n, j->off + (off_t)done);
            if (g < 0) { j->got = -1; break; }
            if ((size_t)g < n) memset(cb + g, 0, n - (size_t)g);
            if (j->xs > 1) bswap_copy(cb, cb, n / j->xs, j->xs);
            convert(j->mem + done / j->xs * j->ms, j->memtype, cb, j->xtype, n / j->xs);
            j->got += g;
        }
        free(cb);
        return NULL;
    }
    if (j->xs == 1) {                                    /* bytes: straight into place */
        j->got = pread_all(j->fd, j->mem, j->bytes, j->off);
        if (j->got >= 0 && (size_t)j->got < j->bytes) memset(j->mem + j->got, 0, j->bytes - (size_t)j->got);
        return NULL;
    }
    /* Through a buffer that stays in the core's cache: the file's bytes are read into it and swapped from it into place, so
     * the destination is written once and never read (swapping in place after one big read moved every byte of it over the
     * memory bus three times: 28 -> 34 GB/s out of the page cache on 8 -> 16 threads was all that bought). */
    enum { kBuf = 256 * 1024 };
    unsigned char *buf = (unsigned char *)malloc(kBuf);
    if (!buf) {                                          /* no buffer: the plain way */
        j->got = pread_all(j->fd, j->mem, j->bytes, j->off);
        if (j->got >= 0) {
            if ((size_t)j->got < j->bytes) memset(j->mem + j->got, 0, j->bytes - (size_t)j->got);
            bswap_copy(j->mem, j->mem, j->bytes / j->xs, j->xs);
        }
        return NULL;
    }
    const size_t step = (size_t)kBuf / j->xs * j->xs;
    size_t done = 0;
    j->got = 0;
    while (done < j->bytes) {
        const size_t n = j->bytes - done < step ? j->bytes - done : step;
        const ssize_t g = pread_all(j->fd, buf, n, j->off + (off_t)done);
        if (g < 0) { j->got = -1; break; }
        if ((size_t)g < n) memset(buf + g, 0, n - (size_t)g);                 /* past the end of the file: zeros, as before */
        bswap_copy(j->mem + done, buf, n / j->xs, j->xs);
        j->got += g;
        done += n;
    }
    free(buf);
    return NULL;
}

static int read_run_parallel(int fd, unsigned char *mem, size_t bytes, size_t xs, off_t off, int xtype, int memtype, size_t ms) {
    enum { kMaxThreads = 32 };
    const size_t kMinPart = (size_t)(xtype != memtype && xs < ms ? 4 : 16) << 20;        /* (file bytes per thread) */
    int nt = (int)(bytes / kMinPart);
    /* default: up to 16 threads, not more than the processors that are online (8 moved 28 GB/s out of the page cache on the
     * 128-core host of the GPU box -- less than the 57 GB/s the copy to the GPU takes them at); CDFT_IO_THREADS = 1 .. 32 */
    int cap = 16;
    const long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (ncpu >= 1 && ncpu < cap) cap = (int)ncpu;
    const char *env = getenv("CDFT_IO_THREADS");
    if (env && *env) { const int e = atoi(env); if (e >= 1) cap = e < kMaxThreads ? e : kMaxThreads; }
    if (nt > cap) nt = cap;
    if (nt < 2) {
        ReadJob j = {fd, mem, bytes, xs, off, 0, xtype, memtype, ms};
        read_job(&j);
        return j.got < 0 ? NCC_EIO : 0;
    }
    ReadJob jobs[kMaxThreads];
    pthread_t th[kMaxThreads];
    int started[kMaxThreads] = {0};
    const size_t nel = bytes / xs, per = (nel + (size_t)nt - 1) / (size_t)nt;
    int rc = 0;
    for (int t = 0; t < nt; ++t) {
        const size_t e0 = per * (size_t)t, e1 = e0 + per < nel ? e0 + per : nel;
        jobs[t] = (ReadJob){fd, mem + e0 * (xtype != memtype ? ms : xs), e0 < e1 ? (e1 - e0) * xs : 0, xs, off + (off_t)(e0 * xs), 0, xtype, memtype, ms};
        if (jobs[t].bytes == 0) continue;
        if (t == nt - 1 || pthread_create(&th[t], NULL, read_job, &jobs[t]) != 0) read_job(&jobs[t]);   /* inline */
        else started[t] = 1;
    }
    for (int t = 0; t < nt; ++t) {
        if (started[t]) pthread_join(th[t], NULL);
        if (jobs[t].got < 0) rc = NCC_EIO;
    }
    return rc;
}

/* The same for writes of the external type: each thread swaps its part through a cache-resident buffer and writes it at its
 * place (before: one thread, a temporary as large as the slab allocated and faulted in at every call, 1.6 GB/s into the page
 * cache -- 12 of the 13.6 s cdfvT took on an eORCA12 record were its four 4.7 GB outputs). */
typedef struct { int fd; const unsigned char *mem; size_t bytes, xs; off_t off; int rc; int xtype, memtype; size_t ms; } WriteJob;   /* bytes: of the file */

static int pwrite_all(int fd, const void *buf, size_t n, off_t off) {
    size_t put = 0;                                      /* (a single pwrite stops near 2 GiB) */
    while (put < n) {
        const ssize_t g = pwrite(fd, (const char *)buf + put, n - put, off + (off_t)put);
        if (g < 0 && errno == EINTR) continue;
        if (g <= 0) return -1;
        put += (size_t)g;
    }
    return 0;
}

static void *write_job(void *arg) {
    WriteJob *j = (WriteJob *)arg;
    j->rc = 0;
    if (j->xtype != j->memtype) {                        /* convert into the buffer, swap there, write (r8 results into float variables) */
        enum { kCBuf = 256 * 1024 };
        unsigned char *cb = (unsigned char *)malloc(kCBuf);
        if (!cb) { j->rc = NCC_ENOMEM; return NULL; }
        const size_t step = (size_t)kCBuf / j->xs * j->xs;
        for (size_t done = 0; done < j->bytes; done += step) {
            const size_t n = j->bytes - done < step ? j->bytes - done : step;
            convert(cb, j->xtype, j->mem + done / j->xs * j->ms, j->memtype, n / j->xs);
            if (j->xs > 1) bswap_copy(cb, cb, n / j->xs, j->xs);
            if (pwrite_all(j->fd, cb, n, j->off + (off_t)done)) { j->rc = NCC_EIO; break; }
        }
        free(cb);
        return NULL;
    }
    if (j->xs == 1) { if (pwrite_all(j->fd, j->mem, j->bytes, j->off)) j->rc = NCC_EIO; return NULL; }
    enum { kBuf = 256 * 1024 };
    unsigned char *buf = (unsigned char *)malloc(kBuf);
    if (!buf) { j->rc = NCC_ENOMEM; return NULL; }
    const size_t step = (size_t)kBuf / j->xs * j->xs;
    for (size_t done = 0; done < j->bytes; done += step) {
        const size_t n = j->bytes - done < step ? j->bytes - done : step;
        bswap_copy(buf, j->mem + done, n / j->xs, j->xs);
        if (pwrite_all(j->fd, buf, n, j->off + (off_t)done)) { j->rc = NCC_EIO; break; }
    }
    free(buf);
    return NULL;
}

static int write_run_parallel(int fd, const unsigned char *mem, size_t bytes, size_t xs, off_t off, int xtype, int memtype, size_t ms) {
    enum { kMaxThreads = 32 };
    const size_t kMinPart = (size_t)8 << 20;
    int nt = (int)(bytes / kMinPart);
    int cap = 16;
    const long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (ncpu >= 1 && ncpu < cap) cap = (int)ncpu;
    const char *env = getenv("CDFT_IO_THREADS");
    if (env && *env) { const int e = atoi(env); if (e >= 1) cap = e < kMaxThreads ? e : kMaxThreads; }
    if (nt > cap) nt = cap;
    if (nt < 2) {
        WriteJob j = {fd, mem, bytes, xs, off, 0, xtype, memtype, ms};
        write_job(&j);
        return j.rc;
    }
    WriteJob jobs[kMaxThreads];
    pthread_t th[kMaxThreads];
    int started[kMaxThreads] = {0};
    const size_t nel = bytes / xs, per = (nel + (size_t)nt - 1) / (size_t)nt;
    int rc = 0;
    for (int t = 0; t < nt; ++t) {
        const size_t e0 = per * (size_t)t, e1 = e0 + per < nel ? e0 + per : nel;
        jobs[t] = (WriteJob){fd, mem + e0 * (xtype != memtype ? ms : xs), e0 < e1 ? (e1 - e0) * xs : 0, xs, off + (off_t)(e0 * xs), 0, xtype, memtype, ms};
        if (jobs[t].bytes == 0) continue;
        if (t == nt - 1 || pthread_create(&th[t], NULL, write_job, &jobs[t]) != 0) write_job(&jobs[t]);   /* inline */
        else started[t] = 1;
    }
    for (int t = 0; t < nt; ++t) {
        if (started[t]) pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    return rc;
}

static int vara_f(File *f, int varid, const size_t *start, const size_t *count, int memtype, void *data, int writing) {
    if (f->indef) return NCC_EINDEFINE;
    if (writing && !f->writable) return NCC_EPERM;
    if (varid < 0 || varid >= f->nvars) return NCC_ENOTVAR;
    Var *v = &f->vars[varid];
    const size_t xs = tsize(v->type), ms = tsize(memtype);
    if (!ms) return NCC_EBADTYPE;
    if ((v->type == NCC_CHAR) != (memtype == NCC_CHAR)) return NCC_ECHAR;
    const int nd = v->ndims;
    uint64_t shape[NCC_MAX_DIMS + 1], st[NCC_MAX_DIMS + 1], ct[NCC_MAX_DIMS + 1];
    uint64_t total = 1;
    for (int d = 0; d < nd; ++d) {
        st[d] = start ? start[d] : 0;
        const int isrec = (d == 0 && v->isrec);
        shape[d] = isrec ? f->numrecs : f->dims[v->dimids[d]].len;
        ct[d] = count ? count[d] : (shape[d] > st[d] ? shape[d] - st[d] : 0);
        if (!isrec || !writing) {
            if (st[d] > shape[d] || (ct[d] && st[d] >= shape[d])) return NCC_EINVALCOORDS;
            if (st[d] + ct[d] > shape[d]) return NCC_EEDGE;
        }
        total *= ct[d];
    }
    if (total == 0) return 0;
    if (f->nc4w) {
        if (!f->h5) return NCC_EINDEFINE;
        const int rc4 = writing ? g_h5_put(f->h5, varid, nd, st, ct, memtype, data) : g_h5_get(f->h5, varid, nd, st, ct, memtype, data);
        if (!rc4 && writing && v->isrec && st[0] + ct[0] > f->numrecs) f->numrecs = st[0] + ct[0];
        return rc4;
    }
    if (f->h5) return writing ? NCC_EPERM : g_h5_get(f->h5, varid, nd, st, ct, memtype, data);
    /* byte strides of the file layout */
    uint64_t stride[NCC_MAX_DIMS + 1];
    uint64_t s = xs;
    for (int d = nd - 1; d >= 0; --d) {
        if (d == 0 && v->isrec) stride[d] = f->recsize;
        else { stride[d] = s; s *= f->dims[v->dimids[d]].len; }
    }
    /* merge the fastest dimensions that are read whole into one contiguous run */
    int last = nd;                         /* first dimension folded into the run (nd: none yet) */
    uint64_t run = 1;
    while (last > 0) {
        const int d = last - 1;
        if (d == 0 && v->isrec) break;                                        /* records are not contiguous */
        if (d < nd - 1 && !(st[d + 1] == 0 && ct[d + 1] == shape[d + 1])) break; /* inner dim not read whole */
        run *= ct[d];
        last = d;
    }
    /* dims 0..last-1 are iterated, dims last..nd-1 form one contiguous run */
    const int nouter = last;
    uint64_t idx[NCC_MAX_DIMS + 1] = {0};
    const size_t tmpn = (size_t)run * xs;
    /* a conversion (v->type != memtype, other than between the two byte types): done by the threads that read or write, each
     * through its cache-resident buffer -- no slab-sized temporaries (there were two, allocated and faulted in at every call) */
    const int xt = (xs == 1 && ms == 1) ? memtype : v->type;
    unsigned char *mem = (unsigned char *)data;
    int rc = 0;
    for (;;) {
        uint64_t off = v->begin + (last < nd ? st[last] * stride[last] : 0);
        for (int d = 0; d < nouter; ++d) off += (st[d] + idx[d]) * stride[d];
        if (writing) rc = write_run_parallel(f->fd, mem, tmpn, xs, (off_t)off, xt, memtype, ms);   /* swapped [converted] and written by parts */
        else         rc = read_run_parallel(f->fd, mem, tmpn, xs, (off_t)off, xt, memtype, ms);    /* a never-written tail reads as 0 */
        if (rc) break;
        mem += (size_t)run * ms;
        int d = nouter - 1;
        while (d >= 0 && ++idx[d] == ct[d]) { idx[d] = 0; --d; }
        if (d < 0) break;
    }
    if (!rc && writing) {
        f->has_data = 1;
        if (v->isrec && st[0] + ct[0] > f->numrecs) { f->numrecs = st[0] + ct[0]; f->nrec_dirty = 1; }
    }
    return rc;
}

static int vara(int ncid, int varid, const size_t *start, const size_t *count, int memtype, void *data, int writing) {
    File *f = getf(ncid);
    if (!f) return NCC_EBADID;
    return vara_f(f, varid, start, count, memtype, data, writing);
}

int ncc_put_vara(int ncid, int varid, const size_t *start, const size_t *count, int memtype, const void *data) {
    return vara(ncid, varid, start, count, memtype, (void *)data, 1);
}

static void *behind_worker(void *arg) {
    Behind *b = (Behind *)arg;
    pthread_mutex_lock(&b->mu);
    for (;;) {
        while (b->n == 0 && !b->stop) pthread_cond_wait(&b->cv, &b->mu);
        if (b->n == 0 && b->stop) break;
        const BehindReq r = b->q[b->head];
        b->head = (b->head + 1) % kBehindQueue;
        --b->n;
        b->busy = 1;
        pthread_cond_broadcast(&b->cv);                    /* (a slot is free) */
        pthread_mutex_unlock(&b->mu);
        const int rc = vara_f(b->f, r.varid, r.has_start ? r.start : NULL, r.has_count ? r.count : NULL, r.memtype, (void *)r.data, 1);
        pthread_mutex_lock(&b->mu);
        b->busy = 0;
        if (rc && !b->rc) b->rc = rc;                      /* the first error is what ncc_wait_writes returns */
        pthread_cond_broadcast(&b->cv);
    }
    pthread_mutex_unlock(&b->mu);
    return NULL;
}

static void behind_stop(File *f) {                          /* (the queue is empty: every caller came through getf) */
    Behind *b = f->wb;
    if (!b) return;
    pthread_mutex_lock(&b->mu);
    b->stop = 1;
    pthread_cond_broadcast(&b->cv);
    pthread_mutex_unlock(&b->mu);
    pthread_join(b->th, NULL);
    pthread_mutex_destroy(&b->mu);
    pthread_cond_destroy(&b->cv);
    free(b);
    f->wb = NULL;
}

/* ncc_put_vara that returns before the data is in the file: see "Write-behind" above.  Errors come back from ncc_wait_writes
 * (NetCDF-4 output and files in define mode are written at once, as by ncc_put_vara). */
int ncc_put_vara_behind(int ncid, int varid, const size_t *start, const size_t *count, int memtype, const void *data) {
    File *f = getf_nowait(ncid);
    if (!f) return NCC_EBADID;
    if (f->nc4w || f->h5 || f->indef || !f->writable || varid < 0 || varid >= f->nvars)
        return vara(ncid, varid, start, count, memtype, (void *)data, 1);
    if (!f->wb) {
        Behind *b = (Behind *)calloc(1, sizeof(Behind));
        if (!b) return NCC_ENOMEM;
        b->f = f;
        pthread_mutex_init(&b->mu, NULL);
        pthread_cond_init(&b->cv, NULL);
        if (pthread_create(&b->th, NULL, behind_worker, b) != 0) {       /* no thread: at once */
            pthread_mutex_destroy(&b->mu); pthread_cond_destroy(&b->cv); free(b);
            return vara(ncid, varid, start, count, memtype, (void *)data, 1);
        }
        f->wb = b;
    }
    Behind *b = f->wb;
    BehindReq r;
    memset(&r, 0, sizeof r);
    r.varid = varid; r.memtype = memtype; r.data = data;
    const int nd = f->vars[varid].ndims;
    if (start) { r.has_start = 1; for (int d = 0; d < nd; ++d) r.start[d] = start[d]; }
    if (count) { r.has_count = 1; for (int d = 0; d < nd; ++d) r.count[d] = count[d]; }
    pthread_mutex_lock(&b->mu);
    while (b->n == kBehindQueue) pthread_cond_wait(&b->cv, &b->mu);
    b->q[(b->head + b->n) % kBehindQueue] = r;
    ++b->n;
    pthread_cond_broadcast(&b->cv);
    pthread_mutex_unlock(&b->mu);
    return 0;
}

int ncc_wait_writes(int ncid) {
    File *f = getf(ncid);                                   /* (waits) */
    if (!f) return NCC_EBADID;
    if (!f->wb) return 0;
    const int rc = f->wb->rc;
    f->wb->rc = 0;
    return rc;
}

int ncc_get_vara(int ncid, int varid, const size_t *start, const size_t *count, int memtype, void *data) {
    return vara(ncid, varid, start, count, memtype, data, 0);
}

const char *ncc_strerror(int status) {
    switch (status) {
        case NCC_NOERR: return "No error";
        case NCC_EBADID: return "NetCDF: Not a valid ID";
        case NCC_EEXIST: return "NetCDF: File exists && NC_NOCLOBBER";
        case NCC_EINVAL: return "NetCDF: Invalid argument";
        case NCC_EPERM: return "NetCDF: Write to read only";
        case NCC_ENOTINDEFINE: return "NetCDF: Operation not allowed in data mode";
        case NCC_EINDEFINE: return "NetCDF: Operation not allowed in define mode";
        case NCC_EINVALCOORDS: return "NetCDF: Index exceeds dimension bound";
        case NCC_ENAMEINUSE: return "NetCDF: String match to name in use";
        case NCC_ENOTATT: return "NetCDF: Attribute not found";
        case NCC_EBADTYPE: return "NetCDF: Not a valid data type or _FillValue type mismatch";
        case NCC_EBADDIM: return "NetCDF: Invalid dimension ID or name";
        case NCC_EUNLIMIT: return "NetCDF: NC_UNLIMITED size already in use";
        case NCC_ENOTVAR: return "NetCDF: Variable not found";
        case NCC_ENOTNC: return "NetCDF: Unknown file format (not CDF-1/2/5 and not NetCDF-4/HDF5)";
        case NCC_ECHAR: return "NetCDF: Attempt to convert between text & numbers";
        case NCC_EEDGE: return "NetCDF: Start+count exceeds dimension bound";
        case NCC_ERANGE: return "NetCDF: Numeric conversion not representable";
        case NCC_ENOMEM: return "NetCDF: Memory allocation (malloc) failure";
        case NCC_EVARSIZE: return "NetCDF: One or more variable sizes violate format constraints";
        case NCC_EIO: return "NetCDF: I/O failure";
        case NCC_ENOENT: return "No such file or directory";
        case NCC_EHDFERR: return g_h5_msg;
        default: return "NetCDF: Unknown Error";
    }
}
