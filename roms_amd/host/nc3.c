/* nc3.c -- the NetCDF "classic" on-disk format, 64-bit-offset variant (CDF-2), written and read directly.
 *
 * The reference writes its history and restart files through the NetCDF library with the NF90_64BIT_OFFSET
 * creation mode (ROMS/Utility/mod_netcdf.F: netcdf_create, CMODE = nf90_64bit_offset unless HDF5/PARALLEL_IO).
 * That library is not part of this image, and the format is small and published (NetCDF User's Guide, "File
 * Format Specification", classic format): a big-endian header -- magic "CDF\2", number of records, dimension
 * list, global attributes, variable list (name, dimension ids, attributes, type, size, 64-bit begin offset) --
 * followed by the fixed-size variables in definition order and then the records, each the record variables'
 * slabs in definition order.  This file implements exactly that: enough to produce files any NetCDF reader
 * opens (tests read them back with scipy.io.netcdf_file, an independent implementation) and to read them again
 * for restart.  Types used: NC_CHAR, NC_INT, NC_DOUBLE.  No fill mode: every variable is written in full
 * (the reference runs with nf90_nofill, mod_netcdf.F: netcdf_enddef's set_fill).
 *
 * C ABI, called from the Fortran host (roms_output.f90) through ISO_C_BINDING.  Handles are small integers.
 * Every function returns 0 on success (the reference's nf90_noerr) or a negative code.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { NC_CHAR = 2, NC_INT = 4, NC_DOUBLE = 6 };
enum { TAG_DIM = 10, TAG_VAR = 11, TAG_ATT = 12 };
#define NC3_MAXDIM 32
#define NC3_MAXVAR 160
#define NC3_MAXATT 24
#define NC3_MAXFILE 8
#define NC3_NAME 64

typedef struct { char name[NC3_NAME]; int type; long n; void *val; } nc_att;       /* val: n elements, host order */
typedef struct { char name[NC3_NAME]; long len; } nc_dim;                           /* len 0 = the record dimension */
typedef struct {
  char name[NC3_NAME];
  int type, ndims, dimid[6], natt, isrec;
  nc_att att[NC3_MAXATT];
  long long vsize, begin;        /* bytes per record (or in total), padded to 4; file offset */
} nc_var;
typedef struct {
  FILE *fp;
  int used, writing, defining;
  int ndim, nvar, ngatt, recdim;
  long numrecs;
  long long recsize, recstart;
  nc_dim dim[NC3_MAXDIM];
  nc_var var[NC3_MAXVAR];
  nc_att gatt[NC3_MAXATT];
} nc_file;

static nc_file files[NC3_MAXFILE];

static int tsize(int type) { return type == NC_DOUBLE ? 8 : type == NC_INT ? 4 : 1; }
static long long pad4(long long n) { return (n + 3) / 4 * 4; }

/* ---- big-endian primitives */
static void put_be(unsigned char *p, const void *v, int size) {
  const unsigned char *s = (const unsigned char *)v;
  for (int k = 0; k < size; k++) p[k] = s[size - 1 - k];       /* host is little-endian (x86-64) */
}
static int w_i32(FILE *fp, int32_t v) { unsigned char b[4]; put_be(b, &v, 4); return fwrite(b, 1, 4, fp) == 4 ? 0 : -1; }
static int w_i64(FILE *fp, int64_t v) { unsigned char b[8]; put_be(b, &v, 8); return fwrite(b, 1, 8, fp) == 8 ? 0 : -1; }
static int r_i32(FILE *fp, int32_t *v) { unsigned char b[4]; if (fread(b, 1, 4, fp) != 4) return -1; put_be((unsigned char *)v, b, 4); return 0; }
static int r_i64(FILE *fp, int64_t *v) { unsigned char b[8]; if (fread(b, 1, 8, fp) != 8) return -1; put_be((unsigned char *)v, b, 8); return 0; }
static int w_name(FILE *fp, const char *s) {
  const long n = (long)strlen(s);
  static const char zero[4] = {0, 0, 0, 0};
  if (w_i32(fp, (int32_t)n)) return -1;
  if (fwrite(s, 1, (size_t)n, fp) != (size_t)n) return -1;
  return fwrite(zero, 1, (size_t)(pad4(n) - n), fp) == (size_t)(pad4(n) - n) ? 0 : -1;
}
static int r_name(FILE *fp, char *s) {
  int32_t n;
  char buf[4];
  if (r_i32(fp, &n) || n < 0 || n >= NC3_NAME) return -1;
  if (fread(s, 1, (size_t)n, fp) != (size_t)n) return -1;
  s[n] = 0;
  return fread(buf, 1, (size_t)(pad4(n) - n), fp) == (size_t)(pad4(n) - n) ? 0 : -1;
}
/* n values of `type` from host order to the file (or back), through a bounded buffer */
static int w_vals(FILE *fp, int type, const void *v, long long n) {
  const int sz = tsize(type);
  unsigned char buf[8192];
  const unsigned char *s = (const unsigned char *)v;
  while (n > 0) {
    const long long m = n < 8192 / sz ? n : 8192 / sz;
    if (sz == 1) memcpy(buf, s, (size_t)m);
    else for (long long k = 0; k < m; k++) put_be(buf + k * sz, s + k * sz, sz);
    if (fwrite(buf, (size_t)sz, (size_t)m, fp) != (size_t)m) return -1;
    s += m * sz; n -= m;
  }
  return 0;
}
static int r_vals(FILE *fp, int type, void *v, long long n) {
  const int sz = tsize(type);
  unsigned char buf[8192];
  unsigned char *d = (unsigned char *)v;
  while (n > 0) {
    const long long m = n < 8192 / sz ? n : 8192 / sz;
    if (fread(buf, (size_t)sz, (size_t)m, fp) != (size_t)m) return -1;
    if (sz == 1) memcpy(d, buf, (size_t)m);
    else for (long long k = 0; k < m; k++) put_be(d + k * sz, buf + k * sz, sz);
    d += m * sz; n -= m;
  }
  return 0;
}
static int w_pad(FILE *fp, long long nbytes) {
  static const char zero[4] = {0, 0, 0, 0};
  const long long p = pad4(nbytes) - nbytes;
  return fwrite(zero, 1, (size_t)p, fp) == (size_t)p ? 0 : -1;
}

static int w_atts(FILE *fp, const nc_att *a, int n) {
  if (n == 0) return w_i32(fp, 0) || w_i32(fp, 0);
  if (w_i32(fp, TAG_ATT) || w_i32(fp, n)) return -1;
  for (int k = 0; k < n; k++) {
    if (w_name(fp, a[k].name) || w_i32(fp, a[k].type) || w_i32(fp, (int32_t)a[k].n)) return -1;
    if (w_vals(fp, a[k].type, a[k].val, a[k].n) || w_pad(fp, a[k].n * tsize(a[k].type))) return -1;
  }
  return 0;
}
static int r_atts(FILE *fp, nc_att *a, int *n) {
  int32_t tag, cnt;
  if (r_i32(fp, &tag) || r_i32(fp, &cnt)) return -1;
  if (tag == 0 && cnt == 0) { *n = 0; return 0; }
  if (tag != TAG_ATT || cnt < 0) return -1;
  if (cnt > NC3_MAXATT) return -10;             /* more attributes than this reader holds: its own code (nc3_open: -11) */
  for (int k = 0; k < cnt; k++) {
    int32_t type, len;
    char padbuf[4];
    if (r_name(fp, a[k].name) || r_i32(fp, &type) || r_i32(fp, &len)) return -1;
    if (len < 0 || type < 1 || type > 6) return -1;       /* a corrupt or foreign header: never a negative allocation */
    a[k].type = type; a[k].n = len;
    a[k].val = calloc((size_t)len + 1, (size_t)tsize(type));
    if (!a[k].val || r_vals(fp, type, a[k].val, len)) return -1;
    const long long p = pad4((long long)len * tsize(type)) - (long long)len * tsize(type);
    if (fread(padbuf, 1, (size_t)p, fp) != (size_t)p) return -1;
  }
  *n = cnt;
  return 0;
}
static long long hdr_atts_size(const nc_att *a, int n) {
  long long s = 8;
  for (int k = 0; k < n; k++) s += 4 + pad4((long long)strlen(a[k].name)) + 8 + pad4(a[k].n * tsize(a[k].type));
  return s;
}

static nc_file *get(int h) { return (h >= 0 && h < NC3_MAXFILE && files[h].used) ? &files[h] : NULL; }

static void layout(nc_file *f) {
  long long hs = 4 + 4;                                                      /* magic, numrecs */
  hs += 8;
  for (int d = 0; d < f->ndim; d++) hs += 4 + pad4((long long)strlen(f->dim[d].name)) + 4;
  hs += hdr_atts_size(f->gatt, f->ngatt);
  hs += 8;
  for (int v = 0; v < f->nvar; v++) {
    nc_var *x = &f->var[v];
    hs += 4 + pad4((long long)strlen(x->name)) + 4 + 4 * x->ndims + hdr_atts_size(x->att, x->natt) + 4 + 4 + 8;
    long long n = 1;
    x->isrec = x->ndims > 0 && f->dim[x->dimid[0]].len == 0;
    for (int d = x->isrec ? 1 : 0; d < x->ndims; d++) n *= f->dim[x->dimid[d]].len;
    x->vsize = pad4(n * tsize(x->type));
  }
  long long off = hs;
  for (int v = 0; v < f->nvar; v++) if (!f->var[v].isrec) { f->var[v].begin = off; off += f->var[v].vsize; }
  f->recstart = off;
  f->recsize = 0;
  for (int v = 0; v < f->nvar; v++) if (f->var[v].isrec) { f->var[v].begin = off; off += f->var[v].vsize; f->recsize += f->var[v].vsize; }
}

static int write_header(nc_file *f) {
  FILE *fp = f->fp;
  if (fseeko(fp, 0, SEEK_SET)) return -1;
  if (fwrite("CDF\2", 1, 4, fp) != 4 || w_i32(fp, (int32_t)f->numrecs)) return -1;
  if (f->ndim == 0) { if (w_i32(fp, 0) || w_i32(fp, 0)) return -1; }
  else {
    if (w_i32(fp, TAG_DIM) || w_i32(fp, f->ndim)) return -1;
    for (int d = 0; d < f->ndim; d++) if (w_name(fp, f->dim[d].name) || w_i32(fp, (int32_t)f->dim[d].len)) return -1;
  }
  if (w_atts(fp, f->gatt, f->ngatt)) return -1;
  if (f->nvar == 0) return w_i32(fp, 0) || w_i32(fp, 0);
  if (w_i32(fp, TAG_VAR) || w_i32(fp, f->nvar)) return -1;
  for (int v = 0; v < f->nvar; v++) {
    const nc_var *x = &f->var[v];
    if (w_name(fp, x->name) || w_i32(fp, x->ndims)) return -1;
    for (int d = 0; d < x->ndims; d++) if (w_i32(fp, x->dimid[d])) return -1;
    if (w_atts(fp, x->att, x->natt) || w_i32(fp, x->type)) return -1;
    if (w_i32(fp, x->vsize > 0x7fffffffLL ? -1 : (int32_t)x->vsize) || w_i64(fp, x->begin)) return -1;
  }
  return 0;
}

/* ------------------------------------------------------------------ define mode */
int nc3_create(const char *path, int *h) {
  for (int k = 0; k < NC3_MAXFILE; k++)
    if (!files[k].used) {
      nc_file *f = &files[k];
      memset(f, 0, sizeof(*f));
      f->fp = fopen(path, "w+b");
      if (!f->fp) return -2;
      f->used = 1; f->writing = 1; f->defining = 1; f->recdim = -1;
      *h = k;
      return 0;
    }
  return -3;
}
int nc3_def_dim(int h, const char *name, long len, int *dimid) {
  nc_file *f = get(h);
  if (!f || !f->defining || f->ndim >= NC3_MAXDIM || strlen(name) >= NC3_NAME) return -4;
  if (len == 0) { if (f->recdim >= 0) return -5; f->recdim = f->ndim; }
  strcpy(f->dim[f->ndim].name, name);
  f->dim[f->ndim].len = len;
  *dimid = f->ndim++;
  return 0;
}
/* dimids slowest first, as in the C and CDL convention (the record dimension, if any, first) */
int nc3_def_var(int h, const char *name, int type, int ndims, const int *dimids, int *varid) {
  nc_file *f = get(h);
  if (!f || !f->defining || f->nvar >= NC3_MAXVAR || ndims > 6 || strlen(name) >= NC3_NAME) return -4;
  nc_var *x = &f->var[f->nvar];
  memset(x, 0, sizeof(*x));
  strcpy(x->name, name);
  x->type = type; x->ndims = ndims;
  for (int d = 0; d < ndims; d++) {
    if (dimids[d] < 0 || dimids[d] >= f->ndim) return -4;
    if (d > 0 && f->dim[dimids[d]].len == 0) return -5;
    x->dimid[d] = dimids[d];
  }
  *varid = f->nvar++;
  return 0;
}
static int add_att(nc_file *f, int varid, const char *name, int type, long n, const void *val) {
  nc_att *a;
  if (strlen(name) >= NC3_NAME) return -4;
  if (varid < 0) { if (f->ngatt >= NC3_MAXATT) return -4; a = &f->gatt[f->ngatt++]; }
  else { if (varid >= f->nvar || f->var[varid].natt >= NC3_MAXATT) return -4; a = &f->var[varid].att[f->var[varid].natt++]; }
  strcpy(a->name, name);
  a->type = type; a->n = n;
  a->val = malloc((size_t)(n > 0 ? n : 1) * (size_t)tsize(type));
  if (!a->val) return -6;
  memcpy(a->val, val, (size_t)n * (size_t)tsize(type));
  return 0;
}
/* varid < 0: a global attribute */
int nc3_put_att_text(int h, int varid, const char *name, const char *text) {
  nc_file *f = get(h);
  return (f && f->defining) ? add_att(f, varid, name, NC_CHAR, (long)strlen(text), text) : -4;
}
int nc3_put_att_double(int h, int varid, const char *name, int n, const double *v) {
  nc_file *f = get(h);
  return (f && f->defining) ? add_att(f, varid, name, NC_DOUBLE, n, v) : -4;
}
int nc3_put_att_int(int h, int varid, const char *name, int n, const int *v) {
  nc_file *f = get(h);
  return (f && f->defining) ? add_att(f, varid, name, NC_INT, n, v) : -4;
}
int nc3_enddef(int h) {
  nc_file *f = get(h);
  if (!f || !f->defining) return -4;
  layout(f);
  f->defining = 0;
  return write_header(f);
}

/* ------------------------------------------------------------------ data mode */
static int locate(nc_file *f, int varid, long rec, long long *off, long long *count) {
  if (varid < 0 || varid >= f->nvar) return -4;
  const nc_var *x = &f->var[varid];
  long long n = 1;
  for (int d = x->isrec ? 1 : 0; d < x->ndims; d++) n *= f->dim[x->dimid[d]].len;
  *count = n;
  if (x->isrec) { if (rec < 0) return -7; *off = x->begin + (long long)rec * f->recsize; }
  else *off = x->begin;
  return 0;
}
/* the whole variable (fixed size) or its whole slab of record `rec` (0-based); n = number of elements given */
static int put_var(int h, int varid, long rec, int type, const void *data, long long n) {
  nc_file *f = get(h);
  long long off, count;
  if (!f || f->defining || !f->writing) return -4;
  if (locate(f, varid, rec, &off, &count)) return -7;
  if (f->var[varid].type != type || n != count) return -8;
  if (fseeko(f->fp, (off_t)off, SEEK_SET)) return -1;
  if (w_vals(f->fp, type, data, n) || w_pad(f->fp, n * tsize(type))) return -1;
  if (f->var[varid].isrec && rec + 1 > f->numrecs) f->numrecs = rec + 1;
  return 0;
}
int nc3_put_var_double(int h, int varid, long rec, const double *data, long long n) { return put_var(h, varid, rec, NC_DOUBLE, data, n); }
int nc3_put_var_int(int h, int varid, long rec, const int *data, long long n) { return put_var(h, varid, rec, NC_INT, data, n); }
int nc3_put_var_text(int h, int varid, long rec, const char *data, long long n) { return put_var(h, varid, rec, NC_CHAR, data, n); }

/* numrecs into the header, buffers to disk (the reference: netcdf_sync after every record) */
int nc3_sync(int h) {
  nc_file *f = get(h);
  if (!f || f->defining) return -4;
  if (f->writing) {
    if (fseeko(f->fp, 4, SEEK_SET) || w_i32(f->fp, (int32_t)f->numrecs)) return -1;
  }
  return fflush(f->fp) ? -1 : 0;
}
int nc3_close(int h) {
  nc_file *f = get(h);
  if (!f) return -4;
  int r = 0;
  if (f->writing && !f->defining) r = nc3_sync(h);
  fclose(f->fp);
  for (int k = 0; k < f->ngatt; k++) free(f->gatt[k].val);
  for (int v = 0; v < f->nvar; v++) for (int k = 0; k < f->var[v].natt; k++) free(f->var[v].att[k].val);
  f->used = 0;
  return r;
}

/* ------------------------------------------------------------------ reading */
/* mode 0: read only; 1: read and append/overwrite records of an existing file */
/* a header this reader rejects: close the file and free whatever attribute values were already read.  -9: not a (sound)
   NetCDF-3 file; -11: a sound file with more dimensions, variables or attributes than NC3_MAX* holds */
static int open_fail(nc_file *f, int code) {
  fclose(f->fp);
  for (int k = 0; k < NC3_MAXATT; k++) free(f->gatt[k].val);
  for (int v = 0; v < NC3_MAXVAR; v++) for (int k = 0; k < NC3_MAXATT; k++) free(f->var[v].att[k].val);
  memset(f, 0, sizeof(*f));
  return code;
}
int nc3_open(const char *path, int mode, int *h) {
  for (int k = 0; k < NC3_MAXFILE; k++)
    if (!files[k].used) {
      nc_file *f = &files[k];
      char magic[4];
      int32_t nrec, tag, cnt;
      memset(f, 0, sizeof(*f));
      f->fp = fopen(path, mode ? "r+b" : "rb");
      if (!f->fp) return -2;
      f->recdim = -1;
      if (fread(magic, 1, 4, f->fp) != 4 || memcmp(magic, "CDF", 3) || (magic[3] != 1 && magic[3] != 2)) return open_fail(f, -9);
      const int cdf2 = magic[3] == 2;
      if (r_i32(f->fp, &nrec) || r_i32(f->fp, &tag) || r_i32(f->fp, &cnt)) return open_fail(f, -9);
      f->numrecs = nrec;
      if (tag == TAG_DIM && cnt > NC3_MAXDIM) return open_fail(f, -11);          /* limits of this reader exceeded */
      if (!((tag == TAG_DIM && cnt >= 0) || (tag == 0 && cnt == 0))) return open_fail(f, -9);
      f->ndim = cnt;
      for (int d = 0; d < cnt; d++) {
        int32_t len;
        if (r_name(f->fp, f->dim[d].name) || r_i32(f->fp, &len) || len < 0) return open_fail(f, -9);
        f->dim[d].len = len;
        if (len == 0) f->recdim = d;
      }
      { const int ra = r_atts(f->fp, f->gatt, &f->ngatt); if (ra) return open_fail(f, ra == -10 ? -11 : -9); }
      if (r_i32(f->fp, &tag) || r_i32(f->fp, &cnt)) return open_fail(f, -9);
      if (tag == TAG_VAR && cnt > NC3_MAXVAR) return open_fail(f, -11);
      if (!((tag == TAG_VAR && cnt >= 0) || (tag == 0 && cnt == 0))) return open_fail(f, -9);
      f->nvar = cnt;
      for (int v = 0; v < cnt; v++) {
        nc_var *x = &f->var[v];
        int32_t nd, type, vs;
        if (r_name(f->fp, x->name) || r_i32(f->fp, &nd) || nd < 0 || nd > 6) return open_fail(f, -9);
        x->ndims = nd;
        for (int d = 0; d < nd; d++) { int32_t id; if (r_i32(f->fp, &id) || id < 0 || id >= f->ndim) return open_fail(f, -9); x->dimid[d] = id; }
        { const int ra = r_atts(f->fp, x->att, &x->natt); if (ra) return open_fail(f, ra == -10 ? -11 : -9); }
        if (r_i32(f->fp, &type) || r_i32(f->fp, &vs) || type < 1 || type > 6) return open_fail(f, -9);
        x->type = type;
        if (cdf2) { int64_t b; if (r_i64(f->fp, &b)) return open_fail(f, -9); x->begin = b; }
        else { int32_t b; if (r_i32(f->fp, &b)) return open_fail(f, -9); x->begin = b; }
        x->isrec = nd > 0 && f->dim[x->dimid[0]].len == 0;
        long long n = 1;
        for (int d = x->isrec ? 1 : 0; d < nd; d++) n *= f->dim[x->dimid[d]].len;
        x->vsize = pad4(n * tsize(type));
        if (x->isrec) f->recsize += x->vsize;
      }
      f->used = 1; f->writing = mode != 0; f->defining = 0;
      *h = k;
      return 0;
    }
  return -3;
}
int nc3_inq_nrec(int h, long *n) { nc_file *f = get(h); if (!f) return -4; *n = f->numrecs; return 0; }
int nc3_inq_dimlen(int h, const char *name, long *len) {
  nc_file *f = get(h);
  if (!f) return -4;
  for (int d = 0; d < f->ndim; d++) if (!strcmp(f->dim[d].name, name)) { *len = f->dim[d].len ? f->dim[d].len : f->numrecs; return 0; }
  return -10;
}
int nc3_inq_varid(int h, const char *name, int *varid) {
  nc_file *f = get(h);
  if (!f) return -4;
  for (int v = 0; v < f->nvar; v++) if (!strcmp(f->var[v].name, name)) { *varid = v; return 0; }
  return -10;
}
static int get_var(int h, int varid, long rec, int type, void *data, long long n) {
  nc_file *f = get(h);
  long long off, count;
  if (!f || f->defining) return -4;
  if (locate(f, varid, rec, &off, &count)) return -7;
  if (f->var[varid].type != type || n != count) return -8;
  if (f->var[varid].isrec && rec >= f->numrecs) return -7;
  if (fseeko(f->fp, (off_t)off, SEEK_SET)) return -1;
  return r_vals(f->fp, type, data, n);
}
int nc3_get_var_double(int h, int varid, long rec, double *data, long long n) { return get_var(h, varid, rec, NC_DOUBLE, data, n); }
int nc3_get_var_int(int h, int varid, long rec, int *data, long long n) { return get_var(h, varid, rec, NC_INT, data, n); }
