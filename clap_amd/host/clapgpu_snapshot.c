/*
 * clapgpu_snapshot.c -- reader / writer of the SoA scene snapshot format (include/clapgpu_snapshot.h).
 * Plain C11, no device code: the arrays it carries are handed to the C ABI of include/clapgpu.h.
 */
#include "clapgpu_snapshot.h"
#include "clapgpu.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ALIGN 64u
static const char MAGIC[8] = { 'C', 'L', 'A', 'P', 'S', 'N', 'P', '1' };

struct header { char magic[8]; uint32_t version, n_arrays; uint64_t table_offset, file_bytes; };
struct entry  { char name[CLAPGPU_SNAPSHOT_NAME_MAX]; uint32_t dtype, ndim; uint64_t dims[4]; uint64_t offset; };
_Static_assert(sizeof(struct header) == 32, "header layout");
_Static_assert(sizeof(struct entry) == 96, "table entry layout");

size_t clapgpu_dtype_size(uint32_t dtype)
{
    switch (dtype) {
    case CLAPGPU_DT_U8:  return 1;
    case CLAPGPU_DT_I32: case CLAPGPU_DT_U32: case CLAPGPU_DT_F32: return 4;
    case CLAPGPU_DT_F64: case CLAPGPU_DT_U64: case CLAPGPU_DT_I64: return 8;
    default: return 0;
    }
}

/* product of the dims, 0 on overflow or a bad rank */
static int count_of(uint32_t ndim, const uint64_t *dims, size_t elem, uint64_t *count, uint64_t *bytes)
{
    if (ndim > CLAPGPU_SNAPSHOT_MAX_DIMS || !elem) return 0;
    uint64_t c = 1;
    for (uint32_t d = 0; d < ndim; d++) {
        if (dims[d] && c > UINT64_MAX / dims[d]) return 0;
        c *= dims[d];
    }
    if (c > UINT64_MAX / elem) return 0;
    *count = c;
    *bytes = c * elem;
    return 1;
}

/* ------------------------------------------------------------------ writer */
struct clapgpu_snapshot_writer {
    FILE         *f;
    char         *path;
    struct entry *table;
    uint32_t      n, cap;
    uint64_t      pos;
};

int clapgpu_snapshot_create(clapgpu_snapshot_writer **out, const char *path)
{
    if (!out || !path) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    clapgpu_snapshot_writer *w = calloc(1, sizeof(*w));
    if (!w) return CLAPGPU_ERR_NOMEM;
    w->f = fopen(path, "wb");
    w->path = strdup(path);
    if (!w->f || !w->path) { clapgpu_snapshot_abort(w); return CLAPGPU_ERR_INVALID_ARGUMENTS; }
    struct header h = { 0 };
    if (fwrite(&h, sizeof(h), 1, w->f) != 1) { clapgpu_snapshot_abort(w); return CLAPGPU_ERR_UNKNOWN; }
    w->pos = sizeof(h);
    *out = w;
    return CLAPGPU_OK;
}

static int pad_to(clapgpu_snapshot_writer *w, uint64_t align)
{
    static const char zeros[ALIGN];
    uint64_t pad = (align - w->pos % align) % align;
    if (pad && fwrite(zeros, 1, pad, w->f) != pad) return 0;
    w->pos += pad;
    return 1;
}

int clapgpu_snapshot_add(clapgpu_snapshot_writer *w, const char *name, uint32_t dtype, uint32_t ndim,
                         const uint64_t *dims, const void *data)
{
    uint64_t count, bytes;
    if (!w || !name || !name[0] || strlen(name) >= CLAPGPU_SNAPSHOT_NAME_MAX || (ndim && !dims))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!count_of(ndim, dims, clapgpu_dtype_size(dtype), &count, &bytes) || (bytes && !data))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    for (uint32_t k = 0; k < w->n; k++)
        if (!strcmp(w->table[k].name, name)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (w->n == w->cap) {
        uint32_t cap = w->cap ? 2 * w->cap : 32;
        struct entry *t = realloc(w->table, cap * sizeof(*t));
        if (!t) return CLAPGPU_ERR_NOMEM;
        w->table = t;
        w->cap = cap;
    }
    if (!pad_to(w, ALIGN)) return CLAPGPU_ERR_UNKNOWN;
    struct entry *e = &w->table[w->n];
    memset(e, 0, sizeof(*e));
    strcpy(e->name, name);
    e->dtype = dtype;
    e->ndim = ndim;
    for (uint32_t d = 0; d < ndim; d++) e->dims[d] = dims[d];
    e->offset = w->pos;
    if (bytes && fwrite(data, 1, bytes, w->f) != bytes) return CLAPGPU_ERR_UNKNOWN;
    w->pos += bytes;
    w->n++;
    return CLAPGPU_OK;
}

static void writer_free(clapgpu_snapshot_writer *w)
{
    free(w->table);
    free(w->path);
    free(w);
}

int clapgpu_snapshot_finish(clapgpu_snapshot_writer *w)
{
    if (!w) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    int ok = pad_to(w, ALIGN);
    struct header h;
    memcpy(h.magic, MAGIC, 8);
    h.version = CLAPGPU_SNAPSHOT_VERSION;
    h.n_arrays = w->n;
    h.table_offset = w->pos;
    h.file_bytes = w->pos + (uint64_t)w->n * sizeof(struct entry);
    ok = ok && (!w->n || fwrite(w->table, sizeof(struct entry), w->n, w->f) == w->n);
    ok = ok && fseek(w->f, 0, SEEK_SET) == 0 && fwrite(&h, sizeof(h), 1, w->f) == 1;
    ok = (fclose(w->f) == 0) && ok;
    w->f = NULL;
    if (!ok) remove(w->path);
    writer_free(w);
    return ok ? CLAPGPU_OK : CLAPGPU_ERR_UNKNOWN;
}

void clapgpu_snapshot_abort(clapgpu_snapshot_writer *w)
{
    if (!w) return;
    if (w->f) fclose(w->f);
    if (w->path) remove(w->path);
    writer_free(w);
}

/* ------------------------------------------------------------------ reader */
struct clapgpu_snapshot {
    unsigned char *buf;            /* whole file, 64-byte aligned */
    uint64_t       size;
    struct header  h;
    struct entry  *table;
};

int clapgpu_snapshot_open(clapgpu_snapshot **out, const char *path)
{
    if (!out || !path) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    FILE *f = fopen(path, "rb");
    if (!f) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    int rc = CLAPGPU_ERR_INVALID_ARGUMENTS;
    clapgpu_snapshot *s = calloc(1, sizeof(*s));
    if (!s) { fclose(f); return CLAPGPU_ERR_NOMEM; }
    if (fseek(f, 0, SEEK_END) != 0) goto fail;
    long end = ftell(f);
    if (end < (long)sizeof(struct header) || fseek(f, 0, SEEK_SET) != 0) goto fail;
    s->size = (uint64_t)end;
    s->buf = aligned_alloc(ALIGN, (s->size + ALIGN - 1) / ALIGN * ALIGN);
    if (!s->buf) { rc = CLAPGPU_ERR_NOMEM; goto fail; }
    if (fread(s->buf, 1, s->size, f) != s->size) goto fail;
    memcpy(&s->h, s->buf, sizeof(s->h));
    if (memcmp(s->h.magic, MAGIC, 8) || s->h.version != CLAPGPU_SNAPSHOT_VERSION) goto fail;
    if (s->h.file_bytes != s->size || s->h.table_offset % ALIGN || s->h.table_offset > s->size ||
        (s->size - s->h.table_offset) / sizeof(struct entry) != s->h.n_arrays ||
        (s->size - s->h.table_offset) % sizeof(struct entry))
        goto fail;
    s->table = (struct entry *)(s->buf + s->h.table_offset);
    for (uint32_t k = 0; k < s->h.n_arrays; k++) {
        const struct entry *e = &s->table[k];
        uint64_t count, bytes;
        if (!memchr(e->name, 0, sizeof(e->name)) || !e->name[0]) goto fail;
        if (!count_of(e->ndim, e->dims, clapgpu_dtype_size(e->dtype), &count, &bytes)) goto fail;
        if (e->offset % ALIGN || e->offset < sizeof(struct header) || e->offset > s->h.table_offset ||
            bytes > s->h.table_offset - e->offset)
            goto fail;
        for (uint32_t j = 0; j < k; j++)
            if (!strcmp(s->table[j].name, e->name)) goto fail;
    }
    fclose(f);
    *out = s;
    return CLAPGPU_OK;
fail:
    fclose(f);
    clapgpu_snapshot_close(s);
    return rc;
}

uint32_t clapgpu_snapshot_count(const clapgpu_snapshot *s)
{
    return s ? s->h.n_arrays : 0;
}

int clapgpu_snapshot_at(const clapgpu_snapshot *s, uint32_t index, clapgpu_snapshot_array *out)
{
    if (!s || !out || index >= s->h.n_arrays) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const struct entry *e = &s->table[index];
    uint64_t bytes;
    memset(out, 0, sizeof(*out));
    out->name = e->name;
    out->dtype = e->dtype;
    out->ndim = e->ndim;
    memcpy(out->dims, e->dims, sizeof(out->dims));
    count_of(e->ndim, e->dims, clapgpu_dtype_size(e->dtype), &out->count, &bytes);
    out->data = s->buf + e->offset;
    return CLAPGPU_OK;
}

int clapgpu_snapshot_find(const clapgpu_snapshot *s, const char *name, clapgpu_snapshot_array *out)
{
    if (!s || !name || !out) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    for (uint32_t k = 0; k < s->h.n_arrays; k++)
        if (!strcmp(s->table[k].name, name))
            return clapgpu_snapshot_at(s, k, out);
    return CLAPGPU_ERR_INVALID_ARGUMENTS;
}

void clapgpu_snapshot_close(clapgpu_snapshot *s)
{
    if (!s) return;
    free(s->buf);
    free(s);
}
