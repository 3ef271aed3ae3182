/*
 * ncmodel.h -- the in-memory description of an open file, shared by the classic reader/writer
 * (ncclassic.c) and the NetCDF-4/HDF5 read plug-in (nch5.c).  Internal: not part of the API.
 */
#ifndef NCMODEL_H
#define NCMODEL_H
#include <stdint.h>
#include "ncclassic.h"

typedef struct {
    char name[NCC_MAX_NAME];
    int type;
    size_t nelems;
    void *data; /* native byte order, packed array of `type` */
} Att;

typedef struct {
    char name[NCC_MAX_NAME];
    uint64_t len; /* 0 = unlimited */
} Dim;

typedef struct {
    char name[NCC_MAX_NAME];
    int ndims, dimids[NCC_MAX_DIMS];
    int natts;
    Att *atts;
    int type, isrec;
    uint64_t vsize, begin;
    uint64_t chunk[NCC_MAX_DIMS];   /* NetCDF-4 output: chunk extents (0 = the library's choice) */
    int deflate, shuffle;           /* NetCDF-4 output: deflate level 0..9 and byte shuffle */
} Var;

/* What the HDF5 plug-in (libnch5.so, loaded on the first NetCDF-4 file) exports.  nch5_open fills malloc'ed
 * arrays that the caller frees like its own; *unlim is the id of the (first) unlimited dimension or -1 and
 * *numrecs its current length. */
typedef struct NcH5 NcH5;
typedef int (*nch5_open_fn)(const char *path, NcH5 **h, int *ndims, Dim **dims, int *nvars, Var **vars, int *ngatts,
                            Att **gatts, int *unlim, uint64_t *numrecs);
typedef int (*nch5_get_vara_fn)(NcH5 *h, int varid, int nd, const uint64_t *start, const uint64_t *count, int memtype,
                                void *data);
typedef void (*nch5_close_fn)(NcH5 *h);
/* NetCDF-4 OUTPUT: the file is created from the complete model at the first enddef (dimension scales, variables with
 * their chunking / deflate, attributes); afterwards data are written by hyperslab (the record dimension grows as
 * needed) and attributes may be rewritten; nch5_finish sets the final length of the record dimension and closes. */
typedef int (*nch5_create_fn)(const char *path, NcH5 **h, int ndims, const Dim *dims, int nvars, const Var *vars, int ngatts,
                              const Att *gatts, int unlim);
typedef int (*nch5_put_vara_fn)(NcH5 *h, int varid, int nd, const uint64_t *start, const uint64_t *count, int memtype,
                                const void *data);
typedef int (*nch5_put_atts_fn)(NcH5 *h, int varid, int natts, const Att *atts);     /* varid -1: global */
typedef int (*nch5_finish_fn)(NcH5 *h, uint64_t numrecs);

#endif
