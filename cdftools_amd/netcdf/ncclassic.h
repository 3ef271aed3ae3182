/*
 * ncclassic.h -- a small, self-contained reader/writer of the NetCDF classic formats
 * (CDF-1, CDF-2 "64-bit offset", CDF-5 "64-bit data"), the I/O layer under cdfio.F90.  NetCDF-4 files
 * (HDF5) are READ through an optional plug-in, libnch5.so (nch5.c, needs an HDF5 C library >= 1.10), loaded
 * when the first such file is opened or created; NetCDF-4 files are written whole (ncc_create ... ncc_close), an
 * existing one cannot be opened for writing.
 *
 * Why it exists: neither the build container nor the GPU box has libnetcdf / netcdf-fortran, and
 * the CDFTOOLS programs read and write NetCDF (src/cdfio.F90 is written on the NF90 API).  The
 * Fortran module cdftools_amd/fortran/netcdf.f90 exposes the NF90 subset cdfio needs on top of these
 * functions; where a real netcdf-fortran exists, link it instead (same API).
 *
 * Conventions: C order (slowest dimension first), 0-based ids, varid -1 = global attributes,
 * every function returns 0 (NCC_NOERR) or a negative NCC_E* code (values follow netcdf.h).
 */
#ifndef NCCLASSIC_H
#define NCCLASSIC_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { NCC_BYTE = 1, NCC_CHAR = 2, NCC_SHORT = 3, NCC_INT = 4, NCC_FLOAT = 5, NCC_DOUBLE = 6,
       NCC_UBYTE = 7, NCC_USHORT = 8, NCC_UINT = 9, NCC_INT64 = 10, NCC_UINT64 = 11 };
enum { NCC_FORMAT_CLASSIC = 1, NCC_FORMAT_64BIT_OFFSET = 2, NCC_FORMAT_NETCDF4 = 3, NCC_FORMAT_64BIT_DATA = 5 };
enum { NCC_GLOBAL = -1, NCC_MAX_NAME = 256, NCC_MAX_DIMS = 16 };

#define NCC_NOERR        0
#define NCC_EBADID     (-33)
#define NCC_EEXIST     (-35)
#define NCC_EINVAL     (-36)
#define NCC_EPERM      (-37)
#define NCC_ENOTINDEFINE (-38)
#define NCC_EINDEFINE  (-39)
#define NCC_EINVALCOORDS (-40)
#define NCC_ENAMEINUSE (-42)
#define NCC_ENOTATT    (-43)
#define NCC_EBADTYPE   (-45)
#define NCC_EBADDIM    (-46)
#define NCC_EUNLIMIT   (-54)
#define NCC_ENOTVAR    (-49)
#define NCC_ENOTNC     (-51)
#define NCC_ECHAR      (-56)
#define NCC_EEDGE      (-57)
#define NCC_ERANGE     (-60)
#define NCC_ENOMEM     (-61)
#define NCC_EVARSIZE   (-62)   /* a variable or an offset does not fit the format (CDF-1 / CDF-2 limits) */
#define NCC_EIO        (-68)
#define NCC_ENOENT     (-90)   /* file does not exist / cannot be opened */
#define NCC_EHDFERR   (-101)   /* NetCDF-4 file: HDF5 plug-in missing, or an HDF5 call failed */

int ncc_create(const char *path, int format, int clobber, int *ncid);
int ncc_open(const char *path, int writable, int *ncid);
int ncc_close(int ncid);
int ncc_redef(int ncid);
int ncc_enddef(int ncid);
int ncc_inq_format(int ncid, int *format);

int ncc_def_dim(int ncid, const char *name, size_t len, int *dimid);   /* len 0 = unlimited */
int ncc_def_var(int ncid, const char *name, int xtype, int ndims, const int *dimids, int *varid);

/* NetCDF-4 output (ncc_create with NCC_FORMAT_NETCDF4): chunk extents, slowest dimension first (0 or out of range =
 * chosen by the library), byte shuffle and deflate level 0..9.  No effect on the classic formats. */
int ncc_def_var_chunking(int ncid, int varid, const size_t *chunks);
int ncc_def_var_deflate(int ncid, int varid, int shuffle, int level);

int ncc_inq(int ncid, int *ndims, int *nvars, int *ngatts, int *unlimdimid);
int ncc_inq_dimid(int ncid, const char *name, int *dimid);
int ncc_inq_dim(int ncid, int dimid, char *name, size_t *len);
int ncc_inq_varid(int ncid, const char *name, int *varid);
int ncc_inq_var(int ncid, int varid, char *name, int *xtype, int *ndims, int *dimids, int *natts);

int ncc_inq_att(int ncid, int varid, const char *name, int *xtype, size_t *len);
int ncc_inq_attname(int ncid, int varid, int attnum, char *name);
/* data is in the native memory layout of memtype; numeric types are converted to xtype */
int ncc_put_att(int ncid, int varid, const char *name, int xtype, size_t len, int memtype, const void *data);
int ncc_get_att(int ncid, int varid, const char *name, int memtype, void *data);
int ncc_rename_att(int ncid, int varid, const char *name, const char *newname);
int ncc_copy_att(int ncid_in, int varid_in, const char *name, int ncid_out, int varid_out);

/* hyperslab I/O with conversion between the variable's external type and memtype */
int ncc_put_vara(int ncid, int varid, const size_t *start, const size_t *count, int memtype, const void *data);
/* Write-behind: the write is queued and carried out in order by one thread of the file while the caller goes on (reading other
 * files, computing); data must stay untouched until ncc_wait_writes -- or any other call on this file, every one of which waits
 * for the queue first -- returns.  ncc_wait_writes returns the first error of the writes queued since the last call. */
int ncc_put_vara_behind(int ncid, int varid, const size_t *start, const size_t *count, int memtype, const void *data);
int ncc_wait_writes(int ncid);
int ncc_get_vara(int ncid, int varid, const size_t *start, const size_t *count, int memtype, void *data);

const char *ncc_strerror(int status);

#ifdef __cplusplus
}
#endif
#endif
