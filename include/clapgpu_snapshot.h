/*
 * clapgpu_snapshot.h -- SoA scene snapshot: the on-disk / wire format between a CLAP-side binding
 * and the GPU path.  Part of libclapgpu_scene.so, plain C.
 *
 * A snapshot is a set of named, typed, shaped arrays -- exactly the SoA arrays include/clapgpu.h
 * takes (entities.pos_scale, entities.rot, entities.parent, models.aabb, skeleton.invmx,
 * animation.times, particles.pos, bodies.pos, lights.color ...), so a scene dumped by the engine
 * (from entity3d / model3d / particle_system / phys_body fields; scene.c:1318-1724 is the loader
 * whose result it captures) can be replayed through the kernels without the engine.
 *
 * File layout (little endian):
 *   header   char magic[8] = "CLAPSNP1"; u32 version = 1; u32 n_arrays; u64 table_offset; u64 file_bytes
 *   table    n_arrays x { char name[48]; u32 dtype; u32 ndim; u64 dims[4]; u64 offset }   (96 B each)
 *   payload  every array at a 64-byte aligned offset, C order, no padding inside
 * Returns cerr_enum-compatible ints (error.h:12-49).
 */
#ifndef CLAPGPU_SNAPSHOT_H
#define CLAPGPU_SNAPSHOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLAPGPU_SNAPSHOT_VERSION 1u
#define CLAPGPU_SNAPSHOT_NAME_MAX 48
#define CLAPGPU_SNAPSHOT_MAX_DIMS 4

enum clapgpu_dtype {
    CLAPGPU_DT_U8 = 1, CLAPGPU_DT_I32 = 2, CLAPGPU_DT_U32 = 3, CLAPGPU_DT_F32 = 4,
    CLAPGPU_DT_F64 = 5, CLAPGPU_DT_U64 = 6, CLAPGPU_DT_I64 = 7,
};
size_t clapgpu_dtype_size(uint32_t dtype);           /* 0 for an unknown code */

/* ---- writing ---- */
typedef struct clapgpu_snapshot_writer clapgpu_snapshot_writer;
int clapgpu_snapshot_create(clapgpu_snapshot_writer **out, const char *path);
/* data is copied to the file at once; name must be unique and shorter than 48 bytes */
int clapgpu_snapshot_add(clapgpu_snapshot_writer *w, const char *name, uint32_t dtype, uint32_t ndim,
                         const uint64_t *dims, const void *data);
int clapgpu_snapshot_finish(clapgpu_snapshot_writer *w);          /* writes table + header, closes, frees */
void clapgpu_snapshot_abort(clapgpu_snapshot_writer *w);          /* closes and removes the file */

/* ---- reading ---- */
typedef struct clapgpu_snapshot clapgpu_snapshot;
typedef struct clapgpu_snapshot_array {
    const char *name;
    uint32_t    dtype, ndim;
    uint64_t    dims[CLAPGPU_SNAPSHOT_MAX_DIMS];
    uint64_t    count;             /* product of dims */
    const void *data;              /* 64-byte aligned, valid until clapgpu_snapshot_close() */
} clapgpu_snapshot_array;
/* the whole file is read into memory and validated (magic, version, bounds, sizes, overlaps) */
int clapgpu_snapshot_open(clapgpu_snapshot **out, const char *path);
uint32_t clapgpu_snapshot_count(const clapgpu_snapshot *s);
int clapgpu_snapshot_at(const clapgpu_snapshot *s, uint32_t index, clapgpu_snapshot_array *out);
int clapgpu_snapshot_find(const clapgpu_snapshot *s, const char *name, clapgpu_snapshot_array *out);   /* CERR_NOT_FOUND-like: -2 */
void clapgpu_snapshot_close(clapgpu_snapshot *s);

#ifdef __cplusplus
}
#endif
#endif /* CLAPGPU_SNAPSHOT_H */
