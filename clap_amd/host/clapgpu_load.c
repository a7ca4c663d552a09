/*
 * clapgpu_load.c -- scene.json + glTF -> SoA scene snapshot (include/clapgpu_load.h).
 *
 * Follows, rule by rule, what the engine's loaders read and in which order they create things:
 *   scene_onload              scene.c:1816-1884   top level: "name", "model" [..], "light" [..]
 *   model_new_from_json       scene.c:1318-1724   one model: keys, defaults, mesh choice, entity / character arrays
 *   scene_add_light_from_json scene.c:1726-1813
 *   gltf_json_parse           gltf.c:666-1064     nodes, scenes, buffers, bufferViews, accessors, animations, skins, meshes
 *   gltf_bin_parse            gltf.c:1065-1096    GLB container
 *   gltf_instantiate_one      gltf.c:1158-1331    vertex attributes, skin -> joints / root pose, animations -> channels
 *   model3d_add_skinning      model.c:524-537, animation_add_channel model.c:725-742, light_get light.c:311-340
 * Nothing here is executed per frame; the arithmetic that has to match the engine's bits (euler -> quaternion,
 * mat4x4_invert, mat4x4_from_quat, the mesh AABB) goes through the same helpers as the rest of the library.
 */
#include <ctype.h>
#include <math.h>
#include <stdbool.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "clapgpu.h"
#include "clapgpu_load.h"
#include "clapgpu_scene.h"
#include "clapgpu_snapshot.h"

#define LD_OK            0
#define LD_NOMEM        (-1)        /* CERR_NOMEM */
#define LD_NOT_FOUND    (-2)
#define LD_INVALID      (-3)        /* CERR_INVALID_ARGUMENTS */
#define LD_PARSE        (-4)        /* CERR_PARSE_FAILED */

#define JOINT_TYPE_MAX   6          /* model.h:31-37 */
#define LIGHTS_MAX       128
#define E_VISIBLE        (1u << 0)  /* model.h:294-310 */
#define E_IS_CHARACTER   (1u << 1)
#define E_HAS_PHYSICS    (1u << 4)
#define E_PHYS_IS_BODY   (1u << 5)
#define E_LIGHT_SOURCE   (1u << 8)
#define E_HAS_ARMATURE   (1u << 12)
#define E_IS_ANIMATED    (1u << 13)
#define E_SKIP_CULLING   (1u << 14)
#define E_ALIVE          (1u << 31)

struct ld_err { char *buf; size_t len; };
static int fail(struct ld_err *e, int rc, const char *fmt, ...)
{
    if (e && e->buf && e->len) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(e->buf, e->len, fmt, ap);
        va_end(ap);
    }
    return rc;
}

/* ================================================================================== JSON */
enum { J_NULL, J_BOOL, J_NUMBER, J_STRING, J_ARRAY, J_OBJECT };
struct jnode {
    int           tag;
    char         *key, *str;
    double        num;
    int           b;
    unsigned      count;
    struct jnode *head, *tail, *next;
};

struct jparse { const char *p, *end; int bad; struct jnode **all; size_t n_all, cap_all; };

static struct jnode *jnew(struct jparse *jp, int tag)
{
    struct jnode *n = calloc(1, sizeof(*n));
    if (!n) { jp->bad = 1; return NULL; }
    if (jp->n_all == jp->cap_all) {
        size_t cap = jp->cap_all ? jp->cap_all * 2 : 256;
        struct jnode **a = realloc(jp->all, cap * sizeof(*a));
        if (!a) { free(n); jp->bad = 1; return NULL; }
        jp->all = a; jp->cap_all = cap;
    }
    jp->all[jp->n_all++] = n;
    n->tag = tag;
    return n;
}

static void jfree(struct jparse *jp)
{
    for (size_t i = 0; i < jp->n_all; i++) { free(jp->all[i]->key); free(jp->all[i]->str); free(jp->all[i]); }
    free(jp->all);
    memset(jp, 0, sizeof(*jp));
}

static void jskip(struct jparse *jp) { while (jp->p < jp->end && isspace((unsigned char)*jp->p)) jp->p++; }

static int hex4(const char *s, unsigned *out)
{
    unsigned v = 0;
    for (int i = 0; i < 4; i++) {
        const int c = (unsigned char)s[i];
        if (!isxdigit(c)) return -1;
        v = v * 16 + (unsigned)(isdigit(c) ? c - '0' : tolower(c) - 'a' + 10);
    }
    *out = v;
    return 0;
}

static char *jstring(struct jparse *jp)
{
    if (jp->p >= jp->end || *jp->p != '"') { jp->bad = 1; return NULL; }
    jp->p++;
    size_t cap = 32, n = 0;
    char *s = malloc(cap);
    if (!s) { jp->bad = 1; return NULL; }
    while (jp->p < jp->end && *jp->p != '"') {
        unsigned cp = (unsigned char)*jp->p++;
        if (cp == '\\') {
            if (jp->p >= jp->end) break;
            const char c = *jp->p++;
            switch (c) {
            case 'b': cp = '\b'; break; case 'f': cp = '\f'; break; case 'n': cp = '\n'; break;
            case 'r': cp = '\r'; break; case 't': cp = '\t'; break;
            case 'u':
                if (jp->end - jp->p < 4 || hex4(jp->p, &cp)) { jp->bad = 1; free(s); return NULL; }
                jp->p += 4;
                if (cp >= 0xD800 && cp < 0xDC00 && jp->end - jp->p >= 6 && jp->p[0] == '\\' && jp->p[1] == 'u') {
                    unsigned lo;
                    if (!hex4(jp->p + 2, &lo) && lo >= 0xDC00 && lo < 0xE000) {
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        jp->p += 6;
                    }
                }
                break;
            default: cp = (unsigned char)c; break;            /* \" \\ \/ */
            }
        }
        if (n + 5 > cap) { cap *= 2; char *t = realloc(s, cap); if (!t) { free(s); jp->bad = 1; return NULL; } s = t; }
        if (cp < 0x80) s[n++] = (char)cp;
        else if (cp < 0x800) { s[n++] = (char)(0xC0 | cp >> 6); s[n++] = (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { s[n++] = (char)(0xE0 | cp >> 12); s[n++] = (char)(0x80 | ((cp >> 6) & 0x3F)); s[n++] = (char)(0x80 | (cp & 0x3F)); }
        else { s[n++] = (char)(0xF0 | cp >> 18); s[n++] = (char)(0x80 | ((cp >> 12) & 0x3F)); s[n++] = (char)(0x80 | ((cp >> 6) & 0x3F)); s[n++] = (char)(0x80 | (cp & 0x3F)); }
    }
    if (jp->p >= jp->end) { free(s); jp->bad = 1; return NULL; }
    jp->p++;                                                   /* closing quote */
    s[n] = 0;
    return s;
}

static struct jnode *jvalue(struct jparse *jp, int depth);

static void jappend(struct jnode *parent, struct jnode *child)
{
    if (parent->tail) parent->tail->next = child; else parent->head = child;
    parent->tail = child;
    parent->count++;
}

static struct jnode *jvalue(struct jparse *jp, int depth)
{
    if (depth > 64) { jp->bad = 1; return NULL; }
    jskip(jp);
    if (jp->p >= jp->end) { jp->bad = 1; return NULL; }
    const char c = *jp->p;
    if (c == '{' || c == '[') {
        struct jnode *n = jnew(jp, c == '{' ? J_OBJECT : J_ARRAY);
        if (!n) return NULL;
        const char close = c == '{' ? '}' : ']';
        jp->p++;
        jskip(jp);
        if (jp->p < jp->end && *jp->p == close) { jp->p++; return n; }
        for (;;) {
            char *key = NULL;
            jskip(jp);
            if (c == '{') {
                key = jstring(jp);
                if (!key) return NULL;
                jskip(jp);
                if (jp->p >= jp->end || *jp->p != ':') { free(key); jp->bad = 1; return NULL; }
                jp->p++;
            }
            struct jnode *v = jvalue(jp, depth + 1);
            if (!v) { free(key); return NULL; }
            v->key = key;
            jappend(n, v);
            jskip(jp);
            if (jp->p >= jp->end) { jp->bad = 1; return NULL; }
            if (*jp->p == ',') { jp->p++; continue; }
            if (*jp->p == close) { jp->p++; return n; }
            jp->bad = 1;
            return NULL;
        }
    }
    if (c == '"') {
        struct jnode *n = jnew(jp, J_STRING);
        if (!n) return NULL;
        n->str = jstring(jp);
        return n->str ? n : NULL;
    }
    if ((size_t)(jp->end - jp->p) >= 4 && !strncmp(jp->p, "true", 4)) { struct jnode *n = jnew(jp, J_BOOL); if (n) n->b = 1; jp->p += 4; return n; }
    if ((size_t)(jp->end - jp->p) >= 5 && !strncmp(jp->p, "false", 5)) { struct jnode *n = jnew(jp, J_BOOL); jp->p += 5; return n; }
    if ((size_t)(jp->end - jp->p) >= 4 && !strncmp(jp->p, "null", 4)) { struct jnode *n = jnew(jp, J_NULL); jp->p += 4; return n; }
    if (c == '-' || isdigit((unsigned char)c)) {
        char tmp[64];
        size_t k = 0;
        while (jp->p + k < jp->end && k < sizeof(tmp) - 1 && (isdigit((unsigned char)jp->p[k]) || strchr("+-.eE", jp->p[k]))) k++;
        memcpy(tmp, jp->p, k);
        tmp[k] = 0;
        char *endp;
        const double v = strtod(tmp, &endp);                 /* json.c parses numbers with strtod as well */
        if (endp == tmp) { jp->bad = 1; return NULL; }
        jp->p += endp - tmp;
        struct jnode *n = jnew(jp, J_NUMBER);
        if (n) n->num = v;
        return n;
    }
    jp->bad = 1;
    return NULL;
}

static struct jnode *jdecode(struct jparse *jp, const char *buf, size_t len)
{
    memset(jp, 0, sizeof(*jp));
    jp->p = buf; jp->end = buf + len;
    struct jnode *root = jvalue(jp, 0);
    if (root) { jskip(jp); if (jp->p != jp->end) jp->bad = 1; }
    if (jp->bad || !root) { jfree(jp); return NULL; }
    return root;
}

static struct jnode *jfind(const struct jnode *obj, const char *key)      /* json_find_member: the first match */
{
    if (!obj || obj->tag != J_OBJECT) return NULL;
    for (struct jnode *p = obj->head; p; p = p->next)
        if (p->key && !strcmp(p->key, key)) return p;
    return NULL;
}

/* json_double_array (json.c:1373-1392): every element must be a number; here at most `n` are taken */
static int jdoubles(const struct jnode *arr, double *out, unsigned n)
{
    if (!arr || arr->tag != J_ARRAY) return -1;
    unsigned i = 0;
    for (struct jnode *p = arr->head; p; p = p->next, i++) {
        if (p->tag != J_NUMBER || i >= n) return -1;
        out[i] = p->num;
    }
    return 0;
}

static int *jints_alloc(const struct jnode *arr, unsigned *count)          /* json_int_array_alloc */
{
    if (!arr || arr->tag != J_ARRAY || !arr->count) return NULL;
    int *a = malloc(sizeof(int) * arr->count);
    if (!a) return NULL;
    unsigned i = 0;
    for (struct jnode *p = arr->head; p; p = p->next, i++) {
        if (p->tag != J_NUMBER) { free(a); return NULL; }
        a[i] = (int)p->num;
    }
    *count = arr->count;
    return a;
}

/* ================================================================================== files, base64 */
static int read_file(const char *path, uint8_t **out, size_t *size)
{
    FILE *f = fopen(path, "rb");
    if (!f) return LD_NOT_FOUND;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = n >= 0 ? malloc((size_t)n + 1) : NULL;
    if (!b) { fclose(f); return LD_NOMEM; }
    if (fread(b, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(b); return LD_PARSE; }
    fclose(f);
    b[n] = 0;
    *out = b; *size = (size_t)n;
    return LD_OK;
}

static long b64_decode(uint8_t *dst, size_t cap, const char *src, size_t slen)
{
    unsigned acc = 0, bits = 0;
    size_t n = 0;
    for (size_t i = 0; i < slen; i++) {
        const int c = (unsigned char)src[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=') break;
        else if (isspace(c)) continue;
        else return -1;
        acc = acc << 6 | (unsigned)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            if (n >= cap) return -1;
            dst[n++] = (uint8_t)(acc >> bits);
        }
    }
    return (long)n;
}

/* ================================================================================== glTF */
#define DATA_URI "data:application/octet-stream;base64,"                 /* gltf.c:13 */
enum { PATH_TRANSLATION, PATH_ROTATION, PATH_SCALE, PATH_NONE };         /* model.h chan_path order, gltf.c:131-136 */

struct g_bufview { unsigned buffer; size_t offset, length; };
struct g_accessor { unsigned bufview, comptype, count, comps; size_t offset; };
struct g_node { char *name; float rotation[4], scale[3], translation[3]; int mesh, skin; unsigned id, nr_children; int *ch_arr; };
struct g_skin { const float *invmxs; char *name; int *joints, *nodes; unsigned nr_joints, nr_invmxs; };
struct g_mesh { char *name; int indices, material, POSITION, NORMAL, JOINTS_0, WEIGHTS_0; };
struct g_sampler { int input, output, interp; };
struct g_channel { int sampler, node, path; };
struct g_anim { char *name; struct g_sampler *samplers; unsigned n_samplers; struct g_channel *channels; unsigned n_channels; };

struct gltf {
    uint8_t *file; size_t file_size;
    const uint8_t *bin; size_t bin_size;
    uint8_t **buffers; size_t *buffer_size; unsigned n_buffers;
    struct g_bufview *bufvws; unsigned n_bufvws;
    struct g_accessor *accrs; unsigned n_accrs;
    struct g_node *nodes; unsigned n_nodes;
    struct g_skin *skins; unsigned n_skins;
    struct g_mesh *meshes; unsigned n_meshes;
    struct g_anim *anis; unsigned n_anis;
    int root_node;
};

static void gltf_free(struct gltf *g)
{
    for (unsigned i = 0; i < g->n_buffers; i++) free(g->buffers[i]);
    free(g->buffers); free(g->buffer_size); free(g->bufvws); free(g->accrs);
    for (unsigned i = 0; i < g->n_nodes; i++) { free(g->nodes[i].name); free(g->nodes[i].ch_arr); }
    free(g->nodes);
    for (unsigned i = 0; i < g->n_skins; i++) { free(g->skins[i].name); free(g->skins[i].joints); free(g->skins[i].nodes); }
    free(g->skins);
    for (unsigned i = 0; i < g->n_meshes; i++) free(g->meshes[i].name);
    free(g->meshes);
    for (unsigned i = 0; i < g->n_anis; i++) { free(g->anis[i].name); free(g->anis[i].samplers); free(g->anis[i].channels); }
    free(g->anis);
    free(g->file);
    memset(g, 0, sizeof(*g));
}

static size_t comp_size(unsigned t)                                      /* gltf_type_size, gltf.c:21-50 */
{
    switch (t) {
    case 0x1400: case 0x1401: return 1;
    case 0x1402: case 0x1403: return 2;
    case 0x1404: case 0x1405: case 0x1406: return 4;
    case 0x140a: return 8;
    default: return 0;
    }
}

static unsigned comps_of(const char *type)                               /* data_type_by_name + data_comp_count */
{
    if (!strcasecmp(type, "SCALAR")) return 1;
    if (!strcasecmp(type, "VEC2")) return 2;
    if (!strcasecmp(type, "VEC3")) return 3;
    if (!strcasecmp(type, "VEC4")) return 4;
    if (!strcasecmp(type, "MAT4")) return 16;
    if (!strcasecmp(type, "MAT3")) return 9;
    if (!strcasecmp(type, "MAT2")) return 4;
    return 0;
}

/* gltf_accessor_buf (gltf.c:313-323) with the bounds the engine does not check */
static const void *accr_buf(const struct gltf *g, int accr, size_t *elsz, unsigned *count)
{
    if (accr < 0 || (unsigned)accr >= g->n_accrs) return NULL;
    const struct g_accessor *a = &g->accrs[accr];
    if (a->bufview >= g->n_bufvws) return NULL;
    const struct g_bufview *bv = &g->bufvws[a->bufview];
    if (bv->buffer >= g->n_buffers || !g->buffers[bv->buffer]) return NULL;
    const size_t es = a->comps * comp_size(a->comptype);                /* gltf_accessor_stride: tightly packed */
    const size_t size = g->buffer_size[bv->buffer];
    /* overflow-safe: every term is checked against what is left of the buffer before it is added */
    if (!es || bv->offset > size || a->offset > size - bv->offset) return NULL;
    const size_t room = size - bv->offset - a->offset;
    if (a->count > room / es) return NULL;
    if (elsz) *elsz = es;
    if (count) *count = a->count;
    return g->buffers[bv->buffer] + a->offset + bv->offset;
}

/* A JSON number usable as an index / size / offset: a finite, non-negative integer value below 2^53 (negative, NaN or
 * huge doubles cast to unsigned / size_t are undefined behaviour, and (unsigned)-1 would index wildly). */
static bool jnum_index(const struct jnode *n, double *out)
{
    if (!n || n->tag != J_NUMBER) return false;
    const double v = n->num;
    if (!(v >= 0.0) || !(v < 9007199254740992.0) || v != floor(v)) return false;
    if (out) *out = v;
    return true;
}

static char *jstrdup(const struct jnode *n) { return n && n->tag == J_STRING ? strdup(n->str) : NULL; }
static int jnum_i(const struct jnode *n, int dflt) { return n && n->tag == J_NUMBER ? (int)n->num : dflt; }

static int gltf_json_parse(struct gltf *g, const char *buf, size_t len, struct ld_err *e)
{
    struct jparse jp;
    struct jnode *root = jdecode(&jp, buf, len);
    if (!root) return fail(e, LD_PARSE, "glTF: JSON does not parse");
    int rc = LD_PARSE;
    struct jnode *scenes = jfind(root, "scenes"), *scene = jfind(root, "scene"), *nodes = jfind(root, "nodes"),
                 *mats = jfind(root, "materials"), *meshes = jfind(root, "meshes"), *anis = jfind(root, "animations"),
                 *skins = jfind(root, "skins"), *accrs = jfind(root, "accessors"), *bufvws = jfind(root, "bufferViews"),
                 *bufs = jfind(root, "buffers");
    /* GLTF_CHECK_PROP, gltf.c:703-720: the engine refuses a file without any of these */
#define NEED(n, name, t) if (!(n) || (n)->tag != (t)) { fail(e, LD_PARSE, "glTF: no '%s' property of the expected type", name); goto out; }
    NEED(scenes, "scenes", J_ARRAY); NEED(scene, "scene", J_NUMBER); NEED(nodes, "nodes", J_ARRAY);
    NEED(mats, "materials", J_ARRAY); NEED(meshes, "meshes", J_ARRAY); NEED(accrs, "accessors", J_ARRAY);
    NEED(bufvws, "bufferViews", J_ARRAY); NEED(bufs, "buffers", J_ARRAY);
#undef NEED
    if (anis && anis->tag != J_ARRAY) { fail(e, LD_PARSE, "glTF: 'animations' is not an array"); goto out; }
    g->root_node = -1;

    /* nodes (gltf.c:728-760).  The engine skips nameless nodes and then indexes its array with glTF node numbers:
     * files it reads correctly name every node.  A nameless node is an error here instead of a silent shift. */
    g->nodes = calloc(nodes->count ? nodes->count : 1, sizeof(*g->nodes));
    if (!g->nodes) { rc = LD_NOMEM; goto out; }
    unsigned nid = 0;
    for (struct jnode *n = nodes->head; n; n = n->next, nid++) {
        struct jnode *jname = jfind(n, "name");
        if (n->tag != J_OBJECT || !jname || jname->tag != J_STRING) { fail(e, LD_PARSE, "glTF: node %u has no name (the engine's node table would shift)", nid); goto out; }
        struct g_node *nd = &g->nodes[g->n_nodes++];
        nd->name = strdup(jname->str);
        nd->id = nid;
        nd->mesh = jnum_i(jfind(n, "mesh"), 0);                         /* absent: the zeroed darray slot, i.e. 0 */
        nd->skin = jnum_i(jfind(n, "skin"), 0);
        double d[4];
        struct jnode *j;
        if ((j = jfind(n, "rotation")) && j->tag == J_ARRAY && !jdoubles(j, d, 4)) for (int i = 0; i < 4; i++) nd->rotation[i] = (float)d[i];
        if ((j = jfind(n, "translation")) && j->tag == J_ARRAY && !jdoubles(j, d, 3)) for (int i = 0; i < 3; i++) nd->translation[i] = (float)d[i];
        if ((j = jfind(n, "scale")) && j->tag == J_ARRAY && !jdoubles(j, d, 3)) for (int i = 0; i < 3; i++) nd->scale[i] = (float)d[i];
        if ((j = jfind(n, "children")) && j->tag == J_ARRAY) nd->ch_arr = jints_alloc(j, &nd->nr_children);
    }
    /* scenes (gltf.c:764-795): the first listed node that is not "Light" / "Camera" is the root; later scenes override */
    for (struct jnode *n = scenes->head; n; n = n->next) {
        struct jnode *jname = jfind(n, "name"), *jnodes = jfind(n, "nodes");
        if (n->tag != J_OBJECT || !jname || jname->tag != J_STRING || !jnodes || jnodes->tag != J_ARRAY) continue;
        unsigned cnt = 0;
        int *ids = jints_alloc(jnodes, &cnt);
        for (unsigned i = 0; ids && i < cnt; i++) {
            if (ids[i] < 0 || (unsigned)ids[i] >= g->n_nodes) continue;
            const struct g_node *nd = &g->nodes[ids[i]];
            if (!strcmp(nd->name, "Light") || !strcmp(nd->name, "Camera")) continue;
            g->root_node = ids[i];
            break;
        }
        free(ids);
    }
    /* buffers (gltf.c:798-846) */
    g->buffers = calloc(bufs->count ? bufs->count : 1, sizeof(*g->buffers));
    g->buffer_size = calloc(bufs->count ? bufs->count : 1, sizeof(*g->buffer_size));
    if (!g->buffers || !g->buffer_size) { rc = LD_NOMEM; goto out; }
    for (struct jnode *n = bufs->head; n; n = n->next) {
        struct jnode *jlen = jfind(n, "byteLength"), *juri = jfind(n, "uri");
        if (n->tag != J_OBJECT || !jlen) continue;
        if (!g->n_buffers && g->bin && juri) continue;                  /* the GLB bin buffer has no uri; the others must */
        if ((g->n_buffers || !g->bin) && !juri) continue;
        size_t blen = (size_t)jlen->num;
        uint8_t *b;
        if (juri) {
            const size_t pre = sizeof(DATA_URI) - 1;
            if (juri->tag != J_STRING || strlen(juri->str) < pre || strncmp(juri->str, DATA_URI, pre)) continue;
            const size_t slen = strlen(juri->str) - pre, cap = slen / 4 * 3 + 3;
            if (cap > blen) blen = cap;
            b = calloc(blen ? blen : 1, 1);
            if (!b) { rc = LD_NOMEM; goto out; }
            if (b64_decode(b, blen, juri->str + pre, slen) < 0) { free(b); b = NULL; }   /* a hole keeps the buffer numbering */
        } else {
            if (blen > g->bin_size) { fail(e, LD_PARSE, "glTF: GLB buffer of %zu bytes in a %zu-byte BIN chunk", blen, g->bin_size); goto out; }
            b = malloc(blen ? blen : 1);
            if (!b) { rc = LD_NOMEM; goto out; }
            memcpy(b, g->bin, blen);
        }
        g->buffers[g->n_buffers] = b;
        g->buffer_size[g->n_buffers++] = blen;
    }
    /* bufferViews (gltf.c:849-866): all three members are required by the engine, byteOffset included */
    g->bufvws = calloc(bufvws->count ? bufvws->count : 1, sizeof(*g->bufvws));
    if (!g->bufvws) { rc = LD_NOMEM; goto out; }
    for (struct jnode *n = bufvws->head; n; n = n->next) {
        struct jnode *jbuf = jfind(n, "buffer"), *jlen = jfind(n, "byteLength"), *joff = jfind(n, "byteOffset");
        /* skipped entries do not take a number, as in the engine (gltf.c:857-861); on top of its rules, numbers that are
         * negative, non-finite or fractional are skipped too (the engine would cast them: undefined behaviour) */
        double vbuf, vlen, voff;
        if (!jnum_index(jbuf, &vbuf) || !jnum_index(jlen, &vlen) || !jnum_index(joff, &voff)) continue;
        if (vbuf >= g->n_buffers) continue;
        struct g_bufview *bv = &g->bufvws[g->n_bufvws++];
        bv->buffer = (unsigned)vbuf; bv->offset = (size_t)voff; bv->length = (size_t)vlen;
    }
    /* accessors (gltf.c:869-897) */
    g->accrs = calloc(accrs->count ? accrs->count : 1, sizeof(*g->accrs));
    if (!g->accrs) { rc = LD_NOMEM; goto out; }
    for (struct jnode *n = accrs->head; n; n = n->next) {
        struct jnode *jbv = jfind(n, "bufferView"), *joff = jfind(n, "byteOffset"), *jcount = jfind(n, "count"),
                     *jtype = jfind(n, "type"), *jct = jfind(n, "componentType");
        double vbv, vcount, vct, voff = 0.0;
        if (!jtype || jtype->tag != J_STRING || !jnum_index(jbv, &vbv) || !jnum_index(jcount, &vcount) || !jnum_index(jct, &vct)) continue;
        if (joff && joff->tag == J_NUMBER && !jnum_index(joff, &voff)) continue;
        if (vbv >= g->n_bufvws || vcount > 4294967295.0 || vct > 65535.0) continue;
        const unsigned comps = comps_of(jtype->str);
        if (!comps) continue;
        struct g_accessor *a = &g->accrs[g->n_accrs++];
        a->bufview = (unsigned)vbv; a->comptype = (unsigned)vct; a->count = (unsigned)vcount; a->comps = comps;
        a->offset = (size_t)voff;
    }
    /* animations (gltf.c:491-581) */
    if (anis) {
        g->anis = calloc(anis->count ? anis->count : 1, sizeof(*g->anis));
        if (!g->anis) { rc = LD_NOMEM; goto out; }
        static const char *paths[] = { "translation", "rotation", "scale", "none" };
        static const char *interps[] = { "STEP", "LINEAR", "CUBICSPLINE", "NONE" };
        for (struct jnode *n = anis->head; n; n = n->next) {
            struct jnode *jch = jfind(n, "channels"), *jsm = jfind(n, "samplers");
            if (!jch || jch->tag != J_ARRAY || !jsm || jsm->tag != J_ARRAY) { fail(e, LD_PARSE, "glTF: animation without channels / samplers"); goto out; }
            struct g_anim *an = &g->anis[g->n_anis++];
            an->name = jstrdup(jfind(n, "name"));
            an->channels = calloc(jch->count ? jch->count : 1, sizeof(*an->channels));
            an->samplers = calloc(jsm->count ? jsm->count : 1, sizeof(*an->samplers));
            if (!an->channels || !an->samplers) { rc = LD_NOMEM; goto out; }
            for (struct jnode *c = jch->head; c; c = c->next) {
                struct g_channel *ch = &an->channels[an->n_channels++];
                ch->sampler = -1; ch->node = -1; ch->path = PATH_NONE;
                if (c->tag != J_OBJECT) continue;
                ch->sampler = jnum_i(jfind(c, "sampler"), -1);
                struct jnode *jt = jfind(c, "target");
                if (jt && jt->tag == J_OBJECT) {
                    ch->node = jnum_i(jfind(jt, "node"), -1);
                    struct jnode *jp_ = jfind(jt, "path");
                    if (jp_ && jp_->tag == J_STRING)
                        for (int i = 0; i < 4; i++) if (!strcmp(paths[i], jp_->str)) { ch->path = i; break; }
                }
            }
            for (struct jnode *c = jsm->head; c; c = c->next) {
                struct g_sampler *sm = &an->samplers[an->n_samplers++];
                sm->input = sm->output = sm->interp = -1;
                if (c->tag != J_OBJECT) continue;
                sm->input = jnum_i(jfind(c, "input"), -1);
                sm->output = jnum_i(jfind(c, "output"), -1);
                struct jnode *ji = jfind(c, "interpolation");
                if (ji && ji->tag == J_STRING)
                    for (int i = 0; i < 4; i++) if (!strcmp(interps[i], ji->str)) { sm->interp = i; break; }
            }
        }
    }
    /* skins (gltf.c:583-617).  skin->nodes[] maps a NODE number to its joint and is sized nr_joints by the engine:
     * a joint node numbered >= nr_joints would be written out of bounds there, so it is refused here. */
    if (skins && skins->tag == J_ARRAY) {
        g->skins = calloc(skins->count ? skins->count : 1, sizeof(*g->skins));
        if (!g->skins) { rc = LD_NOMEM; goto out; }
        for (struct jnode *n = skins->head; n; n = n->next) {
            struct g_skin *sk = &g->skins[g->n_skins++];
            struct jnode *jmat = jfind(n, "inverseBindMatrices"), *jj = jfind(n, "joints");
            if (jmat && jmat->tag == J_NUMBER) {
                size_t es; unsigned cnt;
                const void *b = accr_buf(g, (int)jmat->num, &es, &cnt);
                if (!b || es != 64) { fail(e, LD_PARSE, "glTF: inverseBindMatrices accessor is not a readable MAT4 float array"); goto out; }
                sk->invmxs = b; sk->nr_invmxs = cnt;
            }
            sk->name = jstrdup(jfind(n, "name"));
            if (jj && jj->tag == J_ARRAY) {
                sk->joints = jints_alloc(jj, &sk->nr_joints);
                if (!sk->joints) { fail(e, LD_PARSE, "glTF: skin joints are not numbers"); goto out; }
                sk->nodes = malloc(sizeof(int) * sk->nr_joints);
                if (!sk->nodes) { rc = LD_NOMEM; goto out; }
                for (unsigned j = 0; j < sk->nr_joints; j++) sk->nodes[j] = 0;
                for (unsigned j = 0; j < sk->nr_joints; j++) {
                    if (sk->joints[j] < 0 || (unsigned)sk->joints[j] >= sk->nr_joints || (unsigned)sk->joints[j] >= g->n_nodes) {
                        fail(e, LD_PARSE, "glTF: skin joint %u is node %d; the engine's node->joint table holds %u entries", j, sk->joints[j], sk->nr_joints);
                        goto out;
                    }
                    sk->nodes[sk->joints[j]] = (int)j;
                }
            }
        }
    }
    /* meshes (gltf.c:994-1037): the first primitive only; "indices" and "material" are required by the engine */
    g->meshes = calloc(meshes->count ? meshes->count : 1, sizeof(*g->meshes));
    if (!g->meshes) { rc = LD_NOMEM; goto out; }
    for (struct jnode *n = meshes->head; n; n = n->next) {
        struct jnode *jname = jfind(n, "name"), *jprim = jfind(n, "primitives");
        if (!jname || jname->tag != J_STRING || !jprim || jprim->tag != J_ARRAY || !jprim->head) continue;
        jprim = jprim->head;
        struct jnode *jidx = jfind(jprim, "indices"), *jmat = jfind(jprim, "material"), *jattr = jfind(jprim, "attributes");
        if (!jattr || jattr->tag != J_OBJECT || !jidx || !jmat) continue;
        struct g_mesh *m = &g->meshes[g->n_meshes++];
        m->name = strdup(jname->str);
        m->indices = (int)jidx->num; m->material = (int)jmat->num;
        m->POSITION = m->NORMAL = m->JOINTS_0 = m->WEIGHTS_0 = -1;
        for (struct jnode *p = jattr->head; p; p = p->next) {
            if (p->tag != J_NUMBER) continue;
            if (!strcmp(p->key, "POSITION")) m->POSITION = (int)p->num;
            else if (!strcmp(p->key, "NORMAL")) m->NORMAL = (int)p->num;
            else if (!strcmp(p->key, "JOINTS_0")) m->JOINTS_0 = (int)p->num;
            else if (!strcmp(p->key, "WEIGHTS_0")) m->WEIGHTS_0 = (int)p->num;
        }
    }
    rc = LD_OK;
out:
    jfree(&jp);
    return rc;
}

/* gltf_onload (gltf.c:1098-1124): GLB first, plain JSON with data: URIs second */
static int gltf_load_file(struct gltf *g, const char *path, struct ld_err *e)
{
    memset(g, 0, sizeof(*g));
    int rc = read_file(path, &g->file, &g->file_size);
    if (rc) return fail(e, rc, "cannot read '%s'", path);
    struct glb_header { uint32_t magic, version, length; } hdr;
    if (g->file_size >= sizeof(hdr)) {
        memcpy(&hdr, g->file, sizeof(hdr));
        if (hdr.magic == 0x46546C67u && hdr.version >= 2 && hdr.length == g->file_size && g->file_size >= 12 + 8) {
            uint32_t jlen, jtype;
            memcpy(&jlen, g->file + 12, 4); memcpy(&jtype, g->file + 16, 4);
            if (jtype == 0x4E4F534Au && (size_t)12 + 8 + jlen + 8 <= g->file_size) {
                uint32_t blen, btype;
                memcpy(&blen, g->file + 20 + jlen, 4); memcpy(&btype, g->file + 24 + jlen, 4);
                if (btype == 0x004E4942u && (size_t)jlen + blen + 12 + 16 == g->file_size) {
                    g->bin = g->file + 28 + jlen; g->bin_size = blen;
                    rc = gltf_json_parse(g, (const char *)g->file + 20, jlen, e);
                    if (rc) gltf_free(g);
                    return rc;
                }
            }
        }
    }
    rc = gltf_json_parse(g, (const char *)g->file, g->file_size, e);
    if (rc) gltf_free(g);
    return rc;
}

static int gltf_mesh_by_name(const struct gltf *g, const char *name)
{
    for (unsigned i = 0; i < g->n_meshes; i++) if (!strcmp(g->meshes[i].name, name)) return (int)i;
    return -1;
}

/* which mesh model_new_from_json instantiates (scene.c:1391-1419) */
static int gltf_pick_mesh(const struct gltf *g)
{
    if (!g->n_meshes) return -1;
    if (g->n_meshes == 1) return 0;
    const int collision = gltf_mesh_by_name(g, "collision");
    const int root = g->root_node < 0 ? 0 : g->nodes[g->root_node].mesh;     /* gltf_root_mesh, gltf.c:445-454 */
    if (root < 0) {                                                      /* the first mesh that is not the collision mesh */
        for (unsigned i = 0; i < g->n_meshes; i++) if ((int)i != collision) return (int)i;
        return -1;
    }
    return (unsigned)root < g->n_meshes ? root : -1;
}

static int gltf_mesh_skin(const struct gltf *g, int mesh)                 /* gltf.c:456-467 */
{
    if (g->meshes[mesh].JOINTS_0 < 0 || g->meshes[mesh].WEIGHTS_0 < 0) return -1;
    for (unsigned i = 0; i < g->n_nodes; i++)
        if (g->nodes[i].mesh == mesh && g->nodes[i].skin >= 0) return g->nodes[i].skin;
    return -1;
}

/* ================================================================================== one model */
struct ld_anim {
    uint32_t n_channels, n_times, n_data;
    uint32_t *ch_target, *ch_path, *ch_nr, *ch_time_off, *ch_data_off;
    float *times, *data, time_end;
};

struct ld_model {
    char *name;
    float aabb[6];                       /* min xyz, max xyz */
    uint32_t nr_joints;                  /* 0: not skinned */
    int32_t *joint_parent;
    char **joint_name;
    float *invmx, *bind, root_pose[16];
    int32_t joint_types[JOINT_TYPE_MAX];
    uint32_t n_verts;
    float *position, *normal, *weights;
    uint8_t *joints;
    struct ld_anim *anims; uint32_t n_anims;
    char **anim_name;
};

static void model_free(struct ld_model *m)
{
    free(m->name); free(m->joint_parent); free(m->invmx); free(m->bind); free(m->position); free(m->normal);
    free(m->weights); free(m->joints);
    for (uint32_t j = 0; m->joint_name && j < m->nr_joints; j++) free(m->joint_name[j]);
    free(m->joint_name);
    for (uint32_t a = 0; a < m->n_anims; a++) {
        struct ld_anim *an = &m->anims[a];
        free(an->ch_target); free(an->ch_path); free(an->ch_nr); free(an->ch_time_off); free(an->ch_data_off); free(an->times); free(an->data);
        if (m->anim_name) free(m->anim_name[a]);
    }
    free(m->anims); free(m->anim_name);
    memset(m, 0, sizeof(*m));
}

/* vertex_array_aabb_calc (util.c, util.h:133) over tightly packed positions */
static void aabb_calc(float aabb[6], const float *vx, uint32_t n)
{
    aabb[0] = aabb[1] = aabb[2] = INFINITY;
    aabb[3] = aabb[4] = aabb[5] = -INFINITY;
    for (uint32_t i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
            const float v = vx[3 * (size_t)i + j];
            aabb[j] = v < aabb[j] ? v : aabb[j];                        /* min(v, aabb) / max(v, aabb) as util.h's macros evaluate */
            aabb[3 + j] = v > aabb[3 + j] ? v : aabb[3 + j];
        }
}

/* gltf_instantiate_one (gltf.c:1158-1331) without the renderer objects */
static int model_from_gltf(struct ld_model *m, const struct gltf *g, int mesh, int fix_origin, struct ld_err *e)
{
    memset(m, 0, sizeof(*m));
    const struct g_mesh *gm = &g->meshes[mesh];
    m->name = strdup(gm->name);
    size_t es; unsigned cnt;
    const float *vx = accr_buf(g, gm->POSITION, &es, &cnt);
    if (!vx || es != 12) return fail(e, LD_PARSE, "mesh '%s': POSITION is not a readable float VEC3 accessor", gm->name);
    m->n_verts = cnt;
    m->position = malloc((size_t)(cnt ? cnt : 1) * 12);
    if (!m->position) return LD_NOMEM;
    memcpy(m->position, vx, (size_t)cnt * 12);
    aabb_calc(m->aabb, m->position, cnt);                               /* mesh_attr_dup(MESH_VX), mesh.c:128 */
    if (fix_origin) {                                                   /* vertex_array_fix_origin, util.c:77-92 */
        const float c[3] = { (m->aabb[0] + m->aabb[3]) / 2.0f, m->aabb[1], (m->aabb[2] + m->aabb[5]) / 2.0f };
        for (uint32_t i = 0; i < cnt; i++) for (int j = 0; j < 3; j++) m->position[3 * (size_t)i + j] -= c[j];
        aabb_calc(m->aabb, m->position, cnt);
    }
    if (gm->NORMAL >= 0) {
        const float *nx = accr_buf(g, gm->NORMAL, &es, &cnt);
        if (!nx || es != 12 || cnt != m->n_verts) return fail(e, LD_PARSE, "mesh '%s': NORMAL does not match POSITION", gm->name);
        m->normal = malloc((size_t)(cnt ? cnt : 1) * 12);
        if (!m->normal) return LD_NOMEM;
        memcpy(m->normal, nx, (size_t)cnt * 12);
    }
    const int skin = gltf_mesh_skin(g, mesh);
    if (skin < 0 || (unsigned)skin >= g->n_skins) return LD_OK;
    const struct g_skin *s = &g->skins[skin];
    if (!s->nr_joints || !s->invmxs || s->nr_invmxs < s->nr_joints)
        return fail(e, LD_PARSE, "mesh '%s': skin without joints or with fewer inverse bind matrices than joints", gm->name);
    /* vertex joints / weights: mesh_attr_dup widens u8x4 joints to ints (mesh.c:112-121); u8 is kept here, u16 narrowed */
    {
        const struct g_accessor *ja = &g->accrs[gm->JOINTS_0];
        const void *jb = accr_buf(g, gm->JOINTS_0, &es, &cnt);
        if (!jb || ja->comps != 4 || cnt != m->n_verts || (ja->comptype != 0x1401 && ja->comptype != 0x1403))
            return fail(e, LD_PARSE, "mesh '%s': JOINTS_0 is not u8 / u16 VEC4 matching POSITION", gm->name);
        m->joints = malloc((size_t)(cnt ? cnt : 1) * 4);
        if (!m->joints) return LD_NOMEM;
        for (size_t i = 0; i < (size_t)cnt * 4; i++) {
            unsigned v = ((const uint8_t *)jb)[i];
            if (ja->comptype == 0x1403) { uint16_t h; memcpy(&h, (const uint8_t *)jb + 2 * i, 2); v = h; }      /* offsets need not be aligned */
            if (v >= s->nr_joints || v > 255) return fail(e, LD_PARSE, "mesh '%s': vertex joint %u outside the skin's %u joints", gm->name, v, s->nr_joints);
            m->joints[i] = (uint8_t)v;
        }
        const float *wb = accr_buf(g, gm->WEIGHTS_0, &es, &cnt);
        if (!wb || es != 16 || cnt != m->n_verts) return fail(e, LD_PARSE, "mesh '%s': WEIGHTS_0 is not float VEC4 matching POSITION", gm->name);
        m->weights = malloc((size_t)(cnt ? cnt : 1) * 16);
        if (!m->weights) return LD_NOMEM;
        memcpy(m->weights, wb, (size_t)cnt * 16);
    }
    /* model3d_add_skinning (model.c:524-537) */
    const uint32_t J = m->nr_joints = s->nr_joints;
    m->invmx = malloc((size_t)J * 64); m->bind = malloc((size_t)J * 64);
    m->joint_parent = malloc(sizeof(int32_t) * J); m->joint_name = calloc(J, sizeof(char *));
    if (!m->invmx || !m->bind || !m->joint_parent || !m->joint_name) return LD_NOMEM;
    for (int i = 0; i < JOINT_TYPE_MAX; i++) m->joint_types[i] = -1;
    memcpy(m->invmx, s->invmxs, (size_t)J * 64);
    for (uint32_t j = 0; j < J; j++) clapgpu_mat4_invert(m->invmx + 16 * (size_t)j, m->bind + 16 * (size_t)j);
    /* root pose: the node named like the skin (gltf.c:1243-1258) */
    static const float ident[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    memcpy(m->root_pose, ident, sizeof(ident));
    for (unsigned i = 0; i < g->n_nodes && s->name; i++) {
        const struct g_node *nd = &g->nodes[i];
        if (strcmp(nd->name, s->name)) continue;
        const float *r = nd->rotation;
        if (sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]) != 0.0f) {     /* vec4_len(): truthiness only */
            clapgpu_mat4_from_quat(r, m->root_pose);
            m->root_pose[12] = nd->translation[0]; m->root_pose[13] = nd->translation[1];
            m->root_pose[14] = nd->translation[2]; m->root_pose[15] = 1.0f;
        }
        break;
    }
    /* joints: names and children (gltf.c:1263-1274) -> parent links.  gltf_skin_node_to_joint (gltf.c:1150-1156):
     * a node numbered >= nr_joints is no joint (-1: the engine stores that child and never follows it to a real joint) */
    for (uint32_t j = 0; j < J; j++) m->joint_parent[j] = -1;
    for (uint32_t j = 0; j < J; j++) {
        const struct g_node *nd = &g->nodes[s->joints[j]];
        m->joint_name[j] = strdup(nd->name);
        for (unsigned c = 0; c < nd->nr_children; c++) {
            const int cn = nd->ch_arr[c];
            if (cn < 0 || (unsigned)cn >= J) continue;
            const int cj = s->nodes[cn];
            if (cj > 0 && (uint32_t)cj != j) m->joint_parent[cj] = (int32_t)j;     /* joint 0 is where the walk starts (model.c:1583): it has no parent */
        }
    }
    /* animations -> channels (gltf.c:1276-1320, animation_add_channel model.c:725-742) */
    m->anims = calloc(g->n_anis ? g->n_anis : 1, sizeof(*m->anims));
    m->anim_name = calloc(g->n_anis ? g->n_anis : 1, sizeof(char *));
    if (!m->anims || !m->anim_name) return LD_NOMEM;
    for (unsigned a = 0; a < g->n_anis; a++) {
        const struct g_anim *ga = &g->anis[a];
        struct ld_anim an;
        memset(&an, 0, sizeof(an));
        const unsigned nc = ga->n_channels ? ga->n_channels : 1;
        an.ch_target = malloc(4 * nc); an.ch_path = malloc(4 * nc); an.ch_nr = malloc(4 * nc);
        an.ch_time_off = malloc(4 * nc); an.ch_data_off = malloc(4 * nc);
        size_t t_cap = 0, d_cap = 0;
        int bad = !an.ch_target || !an.ch_path || !an.ch_nr || !an.ch_time_off || !an.ch_data_off ? LD_NOMEM : 0;
        for (unsigned c = 0; !bad && c < ga->n_channels; c++) {
            const struct g_channel *ch = &ga->channels[c];
            if (ch->sampler < 0 || (unsigned)ch->sampler >= ga->n_samplers) { bad = fail(e, LD_PARSE, "animation '%s': channel %u has no sampler", ga->name ? ga->name : "", c); break; }
            const struct g_sampler *sm = &ga->samplers[ch->sampler];
            size_t tes, des; unsigned frames, dcnt;
            const float *time = accr_buf(g, sm->input, &tes, &frames);
            const float *data = accr_buf(g, sm->output, &des, &dcnt);
            if (!time || tes != 4 || !data || !frames || dcnt < frames || des % 4)
                { bad = fail(e, LD_PARSE, "animation '%s': channel %u has unreadable key times / values", ga->name ? ga->name : "", c); break; }
            const int joint = ch->node >= 0 && (unsigned)ch->node < J ? s->nodes[ch->node] : -1;      /* gltf_skin_node_to_joint */
            if (joint < 0) continue;                                    /* "references a non-existent joint": skipped */
            const uint32_t dfl = (uint32_t)(des / 4);                   /* floats per key: 3 (T, S) or 4 (R) */
            if (an.n_times + frames > t_cap) { t_cap = (an.n_times + frames) * 2; float *t = realloc(an.times, 4 * t_cap); if (!t) { bad = LD_NOMEM; break; } an.times = t; }
            if (an.n_data + (size_t)frames * dfl > d_cap) { d_cap = (an.n_data + (size_t)frames * dfl) * 2; float *t = realloc(an.data, 4 * d_cap); if (!t) { bad = LD_NOMEM; break; } an.data = t; }
            const uint32_t k = an.n_channels++;
            an.ch_target[k] = (uint32_t)joint; an.ch_path[k] = (uint32_t)ch->path; an.ch_nr[k] = frames;
            an.ch_time_off[k] = an.n_times; an.ch_data_off[k] = an.n_data;
            memcpy(an.times + an.n_times, time, 4 * (size_t)frames);
            memcpy(an.data + an.n_data, data, 4 * (size_t)frames * dfl);
            an.n_times += frames; an.n_data += frames * dfl;
            float last;
            memcpy(&last, (const uint8_t *)time + 4 * (size_t)(frames - 1), 4);
            an.time_end = an.time_end > last ? an.time_end : last;        /* max(an->time_end, time[frames - 1]) */
        }
        if (bad || !an.n_channels) {                                    /* "an animation with no channels has no reason to exist" */
            free(an.ch_target); free(an.ch_path); free(an.ch_nr); free(an.ch_time_off); free(an.ch_data_off); free(an.times); free(an.data);
            if (bad) return bad;
            continue;
        }
        m->anim_name[m->n_anims] = ga->name ? strdup(ga->name) : NULL;
        m->anims[m->n_anims++] = an;
    }
    return LD_OK;
}

/* ================================================================================== snapshot output */
static int add(clapgpu_snapshot_writer *w, const char *comp, const char *key, uint32_t dt, uint32_t nd, uint64_t d0, uint64_t d1, const void *p)
{
    char name[CLAPGPU_SNAPSHOT_NAME_MAX];
    if ((size_t)snprintf(name, sizeof(name), "%s.%s", comp, key) >= sizeof(name)) return LD_INVALID;
    const uint64_t dims[2] = { d0, d1 };
    static const uint64_t zero8[2];
    return clapgpu_snapshot_add(w, name, dt, nd, dims, p ? p : zero8);
}

static int add_i64(clapgpu_snapshot_writer *w, const char *comp, const char *key, int64_t v)
{
    return add(w, comp, key, CLAPGPU_DT_I64, 1, 1, 0, &v);
}

static int write_model(clapgpu_snapshot_writer *w, unsigned k, const struct ld_model *m)
{
    char comp[24], key[40];
    snprintf(comp, sizeof(comp), "model%u", k);
    int rc = 0;
    const uint32_t J = m->nr_joints, V = m->n_verts;
#define A(...) do { if (!rc) rc = add(w, comp, __VA_ARGS__); } while (0)
    if (!rc) rc = add_i64(w, comp, "nr_joints", J);
    if (!rc) rc = add_i64(w, comp, "n_verts", V);
    A("aabb", CLAPGPU_DT_F32, 1, 6, 0, m->aabb);
    A("position", CLAPGPU_DT_F32, 2, V, 3, m->position);
    if (m->normal) A("normal", CLAPGPU_DT_F32, 2, V, 3, m->normal);
    if (J) {
        A("joints", CLAPGPU_DT_U8, 2, V, 4, m->joints);
        A("weights", CLAPGPU_DT_F32, 2, V, 4, m->weights);
        {   /* model.vert:36-38 never renormalises: total_local_pos.w = sum of the weights.  How far this mesh is from 1
             * decides whether a pre-skinned draw may feed vec4(p, 1) (clapgpu_skin_batch.out_w, clapgpu.h) */
            float dev = 0.f;
            for (uint32_t v = 0; v < V; v++) {
                const float *q = m->weights + 4 * (size_t)v;
                float sum = 0.f;
                for (int i = 0; i < 4; i++) sum += q[i];          /* the shader's accumulation order */
                const float d = fabsf(sum - 1.f);
                if (!(d <= dev)) dev = d;                         /* NaN weights surface as NaN */
            }
            A("weight_sum_max_dev", CLAPGPU_DT_F32, 1, 1, 0, &dev);
        }
        A("joint_parent", CLAPGPU_DT_I32, 1, J, 0, m->joint_parent);
        A("invmx", CLAPGPU_DT_F32, 2, J, 16, m->invmx);
        A("bind", CLAPGPU_DT_F32, 2, J, 16, m->bind);
        A("root_pose", CLAPGPU_DT_F32, 1, 16, 0, m->root_pose);
        A("joint_types", CLAPGPU_DT_I32, 1, JOINT_TYPE_MAX, 0, m->joint_types);
        if (!rc) rc = add_i64(w, comp, "n_anims", m->n_anims);
        uint64_t nonstrict_total = 0;
        for (uint32_t a = 0; a < m->n_anims && !rc; a++) {
            const struct ld_anim *an = &m->anims[a];
#define AK(suffix, dt, n, p) do { snprintf(key, sizeof(key), "a%u_%s", a, suffix); A(key, dt, 1, n, 0, p); } while (0)
            AK("ch_target", CLAPGPU_DT_U32, an->n_channels, an->ch_target);
            AK("ch_path", CLAPGPU_DT_U32, an->n_channels, an->ch_path);
            AK("ch_nr", CLAPGPU_DT_U32, an->n_channels, an->ch_nr);
            AK("ch_time_off", CLAPGPU_DT_U32, an->n_channels, an->ch_time_off);
            AK("ch_data_off", CLAPGPU_DT_U32, an->n_channels, an->ch_data_off);
            AK("times", CLAPGPU_DT_F32, an->n_times, an->times);
            AK("data", CLAPGPU_DT_F32, an->n_data, an->data);
            AK("time_end", CLAPGPU_DT_F32, 1, &an->time_end);
            {   /* channel_time_to_idx scans from the cursor joint->off[path] (model.c:1266-1288, 1310): with key times that
                 * do not strictly increase the bracket it finds depends on that cursor's history, which the stateless
                 * device search (pose.hip) does not have.  Flag such channels: count of keys with t[i] <= t[i-1]. */
                uint32_t *ns = calloc(an->n_channels ? an->n_channels : 1, sizeof(*ns));
                if (!ns) { rc = LD_NOMEM; break; }
                uint32_t total = 0;
                for (uint32_t c = 0; c < an->n_channels; c++) {
                    const float *t = an->times + an->ch_time_off[c];
                    for (uint32_t i = 1; i < an->ch_nr[c]; i++)
                        ns[c] += !(t[i] > t[i - 1]);
                    total += ns[c];
                }
                AK("ch_nonstrict", CLAPGPU_DT_U32, an->n_channels, ns);
                free(ns);
                nonstrict_total += total;
            }
#undef AK
        }
        if (!rc) rc = add_i64(w, comp, "key_times_nonstrict", (int64_t)nonstrict_total);
    }
#undef A
    return rc;
}

/* ================================================================================== scene.json */
struct vec { void *p; size_t n, cap, el; };
static void *vpush(struct vec *v)
{
    if (v->n == v->cap) {
        const size_t cap = v->cap ? v->cap * 2 : 64;
        void *p = realloc(v->p, cap * v->el);
        if (!p) return NULL;
        v->p = p; v->cap = cap;
    }
    void *slot = (char *)v->p + v->n++ * v->el;
    memset(slot, 0, v->el);
    return slot;
}

struct ld_entity { float pos_scale[4], rot[4]; int32_t parent, parent_joint, model; uint32_t flags; char *name; int32_t light_idx; };
struct ld_carrier { uint32_t entity; int32_t light; float off[3]; };
struct ld_attach { uint32_t entity, parent, joint; };
struct ld_body { uint32_t entity; int32_t geom_class, phys_type; double mass, radius, length, yoffset, bounce, bounce_vel; };
struct ld_char { uint32_t entity, model; double speed; uint8_t can_jump, can_dash; };

struct ld_lights {
    uint32_t nr_lights;
    float pos[LIGHTS_MAX][3], color[LIGHTS_MAX][3], attenuation[LIGHTS_MAX][3], dir[LIGHTS_MAX][3], cutoff[LIGHTS_MAX];
    int32_t is_dir[LIGHTS_MAX];
    uint32_t active[LIGHTS_MAX];
    float ambient[3], shadow_tint[3];
};

struct ld_scene {
    struct vec models, entities, carriers, attaches, bodies, chars;     /* ld_model, ld_entity, ... */
    struct ld_lights lights;
    const char *asset_dir;
};

static int light_get(struct ld_lights *l)                                /* light.c:311-340 */
{
    int idx = -1;
    for (int i = 0; i < LIGHTS_MAX; i++) if (!l->active[i]) { idx = i; break; }    /* bitmap_set_lowest */
    if (idx < 0) return -1;
    l->active[idx] = 1;
    if ((uint32_t)idx >= l->nr_lights) l->nr_lights = (uint32_t)idx + 1;
    memset(l->pos[idx], 0, 12); memset(l->color[idx], 0, 12); memset(l->dir[idx], 0, 12);
    l->attenuation[idx][0] = 1; l->attenuation[idx][1] = 0; l->attenuation[idx][2] = 0;
    l->cutoff[idx] = 0;
    l->is_dir[idx] = 1;
    return idx;
}

static float to_radians(float degrees) { return (float)(degrees * M_PI / 180.0); }     /* util.h:77-80 */

/* transform_rotate_vec3 for light_update_from_entity's spot direction (light.c:398-400): v' = q v q^-1 via linmath's
 * quat_mul_vec3 (linmath.h: t = 2 cross(q.xyz, v); v' = v + w t + cross(q.xyz, t)) */
static void quat_mul_vec3(float r[3], const float q[4], const float v[3])
{
    float t[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    for (int i = 0; i < 3; i++) t[i] = t[i] * 2.0f;
    const float u[3] = { q[1] * t[2] - q[2] * t[1], q[2] * t[0] - q[0] * t[2], q[0] * t[1] - q[1] * t[0] };
    for (int i = 0; i < 3; i++) { const float wt = t[i] * q[3]; r[i] = v[i] + wt; r[i] = r[i] + u[i]; }
}

static int find_entity(const struct ld_scene *s, const char *name)       /* mq_find_entity: the first entity so named */
{
    const struct ld_entity *e = s->entities.p;
    for (size_t i = 0; i < s->entities.n; i++) if (e[i].name && !strcmp(e[i].name, name)) return (int)i;
    return -1;
}

/* model_new_from_json, scene.c:1318-1724 */
static int model_from_json(struct ld_scene *s, const struct jnode *node, struct ld_err *e)
{
    double mass = 1.0, bounce = 0.0, bounce_vel = INFINITY, geom_off = 0.0, geom_radius = 1.0, geom_length = 1.0, speed = 0.75;
    const char *name = NULL, *gltf = NULL;
    int can_jump = 0, can_dash = 0, fix_origin = 0, geom_class = 0 /* GEOM_SPHERE */, ptype = 0 /* PHYS_BODY */;
    const struct jnode *ent = NULL, *ch = NULL, *phys = NULL;
    if (node->tag != J_OBJECT) return fail(e, LD_PARSE, "scene: model is not an object");
    for (const struct jnode *p = node->head; p; p = p->next) {
        if (p->tag == J_STRING && !strcmp(p->key, "name")) name = p->str;
        else if (p->tag == J_STRING && !strcmp(p->key, "gltf")) gltf = p->str;
        else if (p->tag == J_OBJECT && !strcmp(p->key, "physics")) phys = p;
        else if (p->tag == J_BOOL && !strcmp(p->key, "can_dash")) can_dash = p->b;
        else if (p->tag == J_BOOL && !strcmp(p->key, "can_jump")) can_jump = p->b;
        else if (p->tag == J_ARRAY && !strcmp(p->key, "entity")) ent = p->head;
        else if (p->tag == J_ARRAY && !strcmp(p->key, "character")) ch = p->head;
        else if (p->tag == J_NUMBER && !strcmp(p->key, "speed")) speed = p->num;
        else if (p->tag == J_BOOL && !strcmp(p->key, "fix_origin")) fix_origin = p->b;
    }
    if (!name || !gltf) return fail(e, LD_PARSE, "scene: model without 'name' or 'gltf'");

    char path[4096];
    snprintf(path, sizeof(path), "%s/%s", s->asset_dir, gltf);
    struct gltf g;
    int rc = gltf_load_file(&g, path, e);
    if (rc) return rc;
    const int mesh = gltf_pick_mesh(&g);
    if (mesh < 0) { gltf_free(&g); return fail(e, LD_PARSE, "'%s': no mesh to instantiate", gltf); }
    struct ld_model *m = vpush(&s->models);
    if (!m) { gltf_free(&g); return LD_NOMEM; }
    rc = model_from_gltf(m, &g, mesh, fix_origin, e);
    if (rc) { gltf_free(&g); return rc; }
    free(m->name);
    m->name = strdup(name);                                              /* model3d_set_name */
    const int32_t model_idx = (int32_t)(s->models.n - 1);

    if (phys)
        for (const struct jnode *p = phys->head; p; p = p->next) {
            if (p->tag == J_NUMBER && !strcmp(p->key, "bounce")) bounce = p->num;
            else if (p->tag == J_NUMBER && !strcmp(p->key, "bounce_vel")) bounce_vel = p->num;
            else if (p->tag == J_NUMBER && !strcmp(p->key, "mass")) mass = p->num;
            else if (p->tag == J_NUMBER && !strcmp(p->key, "yoffset")) geom_off = p->num;
            else if (p->tag == J_NUMBER && !strcmp(p->key, "radius")) geom_radius = p->num;
            else if (p->tag == J_NUMBER && !strcmp(p->key, "length")) geom_length = p->num;
            else if (p->tag == J_STRING && !strcmp(p->key, "geom")) {
                if (!strcmp(p->str, "trimesh")) geom_class = 2;          /* physics.h:31-39 geom_class: SPHERE, CAPSULE, TRIMESH */
                else if (!strcmp(p->str, "sphere")) geom_class = 0;
                else if (!strcmp(p->str, "capsule")) geom_class = 1;
            } else if (p->tag == J_STRING && !strcmp(p->key, "type")) {
                if (!strcmp(p->str, "body")) ptype = 0;
                else if (!strcmp(p->str, "geom")) ptype = 1;
            }
        }
    /* "armature": semantic joint roles by joint name (scene.c:1476-1492) */
    const struct jnode *arm = jfind(node, "armature");
    if (arm && arm->tag == J_OBJECT) {
        static const char *jt_str[JOINT_TYPE_MAX] = { NULL, "head", "foot_left", "foot_right", "hand_left", "hand_right" };
        for (int i = 1; i < JOINT_TYPE_MAX; i++) {
            const struct jnode *jj = jfind(arm, jt_str[i]);
            if (!jj || jj->tag != J_STRING) continue;
            for (uint32_t j = 0; j < m->nr_joints; j++)
                if (!strcmp(jj->str, m->joint_name[j])) { m->joint_types[i] = (int32_t)j; break; }
        }
    }
    const int animated = m->nr_joints && m->n_anims;

    for (const struct jnode *it = ent ? ent : ch; it; it = it->next) {
        if (it->tag != J_OBJECT) continue;
        struct ld_entity *en = vpush(&s->entities);
        if (!en) { gltf_free(&g); return LD_NOMEM; }
        m = (struct ld_model *)s->models.p + model_idx;
        const uint32_t ei = (uint32_t)(s->entities.n - 1);
        /* entity3d_make (model.c:1730-1762) */
        en->rot[3] = 1.0f; en->pos_scale[3] = 1.0f;
        en->parent = -1; en->parent_joint = -1; en->light_idx = -1; en->model = model_idx;
        en->flags = E_ALIVE | E_VISIBLE | (m->nr_joints ? E_HAS_ARMATURE : 0) | (m->n_anims ? E_IS_ANIMATED : 0);
        if (ch) {                                                        /* character: skips culling (scene.c:1508-1515) */
            en->flags |= E_IS_CHARACTER | E_SKIP_CULLING;
            struct ld_char *c = vpush(&s->chars);
            if (!c) { gltf_free(&g); return LD_NOMEM; }
            c->entity = ei; c->model = (uint32_t)model_idx; c->speed = speed; c->can_jump = (uint8_t)can_jump; c->can_dash = (uint8_t)can_dash;
        }
        const struct jnode *j;
        if ((j = jfind(it, "name")) && j->tag == J_STRING) en->name = strdup(j->str);
        if ((j = jfind(it, "attach")) && j->tag == J_STRING) {
            const int p = find_entity(s, j->str);
            if (p < 0 || (uint32_t)p == ei) continue;                    /* CRES_RET(mq_find_entity(...), continue) */
            en->parent = p;
        }
        if ((j = jfind(it, "attach_joint")) && j->tag == J_STRING && en->parent >= 0) {
            const struct ld_entity *pe = (struct ld_entity *)s->entities.p + en->parent;
            const struct ld_model *pm = (struct ld_model *)s->models.p + pe->model;
            static const char *jt_str[JOINT_TYPE_MAX] = { NULL, "head", "foot_left", "foot_right", "hand_left", "hand_right" };
            int jt = -1;
            for (int i = 1; i < JOINT_TYPE_MAX; i++) if (!strcmp(jt_str[i], j->str)) jt = i;
            if (jt < 0 || !pm->nr_joints || pm->joint_types[jt] < 0) continue;           /* model3d_joint_by_type fails: continue */
            en->parent_joint = pm->joint_types[jt];
        }
        double d[3] = { 0, 0, 0 };
        if ((j = jfind(it, "rotate")) && j->tag == J_ARRAY && !jdoubles(j, d, 3)) {
            const float ang[3] = { (float)d[0], (float)d[1], (float)d[2] };
            clapgpu_quat_from_angles(ang, 1, en->rot);
        }
        j = jfind(it, "position");
        if (!j || j->tag != J_ARRAY) continue;
        const struct jnode *pos = j->head;
        if (!pos || pos->tag != J_NUMBER) continue;
        const float px = (float)pos->num;
        pos = pos->next;
        if (!pos || pos->tag != J_NUMBER) continue;
        const float py = (float)pos->num;
        pos = pos->next;
        if (!pos || pos->tag != J_NUMBER) continue;
        en->pos_scale[0] = px; en->pos_scale[1] = py; en->pos_scale[2] = (float)pos->num;       /* entity3d_position */
        pos = pos->next;
        if (!pos || pos->tag != J_NUMBER) continue;
        en->pos_scale[3] = (float)pos->num;                              /* entity3d_scale */
        pos = pos->next;
        if (pos && pos->tag == J_NUMBER) {                               /* entity3d_rotate(e, 0, to_radians(deg), 0) */
            const float ang[3] = { 0.0f, to_radians((float)pos->num), 0.0f };
            clapgpu_quat_from_angles(ang, 0, en->rot);
        }
        /* lights carried by the entity (scene.c:1587-1632) */
        struct ld_lights *L = &s->lights;
        float light_off[3] = { 0, 0, 0 };
        if ((j = jfind(it, "light_color")) && j->tag == J_ARRAY) {
            en->light_idx = light_get(L);
            if (en->light_idx < 0) goto light_done;
            if (jdoubles(j, d, 3)) goto light_done;
            for (int i = 0; i < 3; i++) L->color[en->light_idx][i] = (float)d[i];
        }
        if ((j = jfind(it, "light_offset")) && j->tag == J_ARRAY && en->light_idx >= 0) {
            if (jdoubles(j, d, 3)) goto light_done;
            for (int i = 0; i < 3; i++) light_off[i] = (float)d[i];
        }
        if ((j = jfind(it, "light_attenuation")) && j->tag == J_ARRAY && en->light_idx >= 0 && !jdoubles(j, d, 3)) {
            for (int i = 0; i < 3; i++) L->attenuation[en->light_idx][i] = (float)d[i];
            L->is_dir[en->light_idx] = 0;
        }
        if ((j = jfind(it, "light_cutoff")) && j->tag == J_NUMBER && en->light_idx >= 0) {
            L->cutoff[en->light_idx] = to_radians((float)j->num);
            L->is_dir[en->light_idx] = 1;
        }
light_done:
        if (en->light_idx >= 0) {
            en->flags |= E_LIGHT_SOURCE;
            struct ld_carrier *c = vpush(&s->carriers);
            if (!c) { gltf_free(&g); return LD_NOMEM; }
            en = (struct ld_entity *)s->entities.p + ei;
            c->entity = ei; c->light = en->light_idx; memcpy(c->off, light_off, 12);
            for (int i = 0; i < 3; i++) L->pos[en->light_idx][i] = en->pos_scale[i] + light_off[i];      /* light_update_from_entity */
            if (L->is_dir[en->light_idx] && L->cutoff[en->light_idx] > 0.0f) {                        /* spotlight: direction */
                const float fwd[3] = { 0, 0, 1 };
                float dir[3];
                quat_mul_vec3(dir, en->rot, fwd);
                for (int i = 0; i < 3; i++) L->dir[en->light_idx][i] = 0.0f - dir[i];                   /* light_set_direction */
            }
        }
        /* entity3d_add_physics (model.c:1799-1808).  phys_body_new (physics.c:953-985) creates a geom for the capsule and
         * trimesh classes only: with "geom": "sphere" (the default) it returns NULL and the entity stays without physics. */
        if (phys && geom_class != 0) {
            struct ld_body *b = vpush(&s->bodies);
            if (!b) { gltf_free(&g); return LD_NOMEM; }
            b->entity = ei; b->geom_class = geom_class; b->phys_type = ptype; b->mass = mass; b->radius = geom_radius;
            b->length = geom_length; b->yoffset = geom_off; b->bounce = bounce; b->bounce_vel = bounce_vel;
            en->flags |= E_HAS_PHYSICS | (ptype == 0 ? E_PHYS_IS_BODY : 0);      /* phys_body_has_body: type == PHYS_BODY */
        }
        if (en->parent >= 0 && en->parent_joint >= 0) {
            struct ld_attach *a = vpush(&s->attaches);
            if (!a) { gltf_free(&g); return LD_NOMEM; }
            a->entity = ei; a->parent = (uint32_t)en->parent; a->joint = (uint32_t)en->parent_joint;
        }
        (void)animated;
    }
    gltf_free(&g);
    return LD_OK;
}

/* scene_add_light_from_json, scene.c:1726-1813 */
static int light_from_json(struct ld_scene *s, const struct jnode *light, struct ld_err *e)
{
    if (light->tag != J_OBJECT) return fail(e, LD_PARSE, "scene: light is not an object");
    struct ld_lights *L = &s->lights;
    double d[3], c[3];
    const struct jnode *j;
    if ((j = jfind(light, "ambient_color"))) {
        if (j->tag != J_ARRAY || jdoubles(j, d, 3)) return fail(e, LD_PARSE, "scene: bad ambient_color");
        for (int i = 0; i < 3; i++) L->ambient[i] = (float)d[i];
        return LD_OK;
    }
    if ((j = jfind(light, "shadow_tint"))) {
        if (j->tag != J_ARRAY || jdoubles(j, d, 3)) return fail(e, LD_PARSE, "scene: bad shadow_tint");
        for (int i = 0; i < 3; i++) L->shadow_tint[i] = (float)d[i];
        return LD_OK;
    }
    const struct jnode *jpos = jfind(light, "position"), *jcolor = jfind(light, "color");
    if (!jpos || jpos->tag != J_ARRAY || !jcolor || jcolor->tag != J_ARRAY || jdoubles(jpos, d, 3) || jdoubles(jcolor, c, 3))
        return fail(e, LD_PARSE, "scene: light without position / color");
    const int idx = light_get(L);
    if (idx < 0) return fail(e, LD_INVALID, "scene: more than %d lights", LIGHTS_MAX);
    L->is_dir[idx] = 1;
    for (int i = 0; i < 3; i++) { L->pos[idx][i] = (float)d[i]; L->color[idx][i] = (float)c[i]; }
    if ((j = jfind(light, "direction")) && j->tag == J_ARRAY && !jdoubles(j, d, 3))
        for (int i = 0; i < 3; i++) L->dir[idx][i] = 0.0f - (float)d[i];                              /* light_set_direction: 0 - dir */
    if ((j = jfind(light, "attenuation")) && j->tag == J_ARRAY && !jdoubles(j, d, 3)) {
        for (int i = 0; i < 3; i++) L->attenuation[idx][i] = (float)d[i];
        L->is_dir[idx] = 0;
    }
    if ((j = jfind(light, "cutoff")) && j->tag == J_NUMBER) { L->cutoff[idx] = (float)j->num; L->is_dir[idx] = 1; }
    return LD_OK;
}

static void scene_free(struct ld_scene *s)
{
    for (size_t i = 0; i < s->models.n; i++) model_free((struct ld_model *)s->models.p + i);
    for (size_t i = 0; i < s->entities.n; i++) free(((struct ld_entity *)s->entities.p)[i].name);
    free(s->models.p); free(s->entities.p); free(s->carriers.p); free(s->attaches.p); free(s->bodies.p); free(s->chars.p);
}

static int write_scene(const struct ld_scene *s, const char *snapshot_path)
{
    clapgpu_snapshot_writer *w = NULL;
    int rc = clapgpu_snapshot_create(&w, snapshot_path);
    if (rc) return rc;
    const size_t n = s->entities.n, nm = s->models.n;
    const struct ld_entity *en = s->entities.p;
    const struct ld_model *md = s->models.p;
    float *ps = calloc(n ? n : 1, 16), *rot = calloc(n ? n : 1, 16), *maabb = calloc(nm ? nm : 1, 24);
    int32_t *parent = calloc(n ? n : 1, 4), *pj = calloc(n ? n : 1, 4), *model = calloc(n ? n : 1, 4);
    uint32_t *flags = calloc(n ? n : 1, 4), *seqs = calloc(n ? n : 1, 4);
    uint8_t *mskip = calloc(nm ? nm : 1, 1);
    if (!ps || !rot || !maabb || !parent || !pj || !model || !flags || !seqs || !mskip) { rc = LD_NOMEM; goto out; }
    for (size_t i = 0; i < n; i++) {
        memcpy(ps + 4 * i, en[i].pos_scale, 16); memcpy(rot + 4 * i, en[i].rot, 16);
        parent[i] = en[i].parent; pj[i] = en[i].parent_joint; model[i] = en[i].model;
        flags[i] = en[i].flags | CLAPGPU_E_DIRTY | (en[i].parent >= 0 && en[i].parent_joint >= 0 ? CLAPGPU_E_JOINT_ATTACHED : 0);
    }
    for (size_t k = 0; k < nm; k++) memcpy(maabb + 6 * k, md[k].aabb, 24);
#define A(comp, ...) do { if (!rc) rc = add(w, comp, __VA_ARGS__); } while (0)
    if (!rc) rc = add_i64(w, "entities", "n", (int64_t)n);
    A("entities", "pos_scale", CLAPGPU_DT_F32, 2, n, 4, ps);
    A("entities", "rot", CLAPGPU_DT_F32, 2, n, 4, rot);
    A("entities", "parent", CLAPGPU_DT_I32, 1, n, 0, parent);
    A("entities", "parent_joint", CLAPGPU_DT_I32, 1, n, 0, pj);
    A("entities", "model", CLAPGPU_DT_I32, 1, n, 0, model);
    A("entities", "flags", CLAPGPU_DT_U32, 1, n, 0, flags);
    A("entities", "seqs", CLAPGPU_DT_U32, 1, n, 0, seqs);
    A("entities", "model_aabb", CLAPGPU_DT_F32, 2, nm, 6, maabb);
    A("entities", "model_skip", CLAPGPU_DT_U8, 1, nm, 0, mskip);
    if (!rc) rc = add_i64(w, "scene", "n_models", (int64_t)nm);
    for (size_t k = 0; k < nm && !rc; k++) rc = write_model(w, (unsigned)k, &md[k]);
    {   /* lights */
        const struct ld_lights *L = &s->lights;
        if (!rc) rc = add_i64(w, "lights", "nr_lights", L->nr_lights);
        A("lights", "pos", CLAPGPU_DT_F32, 2, LIGHTS_MAX, 3, L->pos);
        A("lights", "color", CLAPGPU_DT_F32, 2, LIGHTS_MAX, 3, L->color);
        A("lights", "attenuation", CLAPGPU_DT_F32, 2, LIGHTS_MAX, 3, L->attenuation);
        A("lights", "dir", CLAPGPU_DT_F32, 2, LIGHTS_MAX, 3, L->dir);
        A("lights", "cutoff", CLAPGPU_DT_F32, 1, LIGHTS_MAX, 0, L->cutoff);
        A("lights", "is_dir", CLAPGPU_DT_I32, 1, LIGHTS_MAX, 0, L->is_dir);
        A("lights", "active", CLAPGPU_DT_U32, 1, LIGHTS_MAX, 0, L->active);
        A("lights", "ambient", CLAPGPU_DT_F32, 1, 3, 0, L->ambient);
        A("lights", "shadow_tint", CLAPGPU_DT_F32, 1, 3, 0, L->shadow_tint);
    }
#define COL(vec, type, field, ctype, dt, comp, key) do { \
        const size_t cn = (vec).n; ctype *col = calloc(cn ? cn : 1, sizeof(ctype)); \
        if (!col) rc = rc ? rc : LD_NOMEM; \
        else { for (size_t q = 0; q < cn; q++) col[q] = (ctype)((const type *)(vec).p)[q].field; \
               A(comp, key, dt, 1, cn, 0, col); free(col); } } while (0)
    COL(s->carriers, struct ld_carrier, entity, uint32_t, CLAPGPU_DT_U32, "carriers", "entity");
    COL(s->carriers, struct ld_carrier, light, int32_t, CLAPGPU_DT_I32, "carriers", "light");
    {
        const size_t cn = s->carriers.n;
        float *off = calloc(cn ? cn : 1, 12);
        if (!off) rc = rc ? rc : LD_NOMEM;
        else { for (size_t q = 0; q < cn; q++) memcpy(off + 3 * q, ((const struct ld_carrier *)s->carriers.p)[q].off, 12);
               A("carriers", "offset", CLAPGPU_DT_F32, 2, cn, 3, off); free(off); }
    }
    COL(s->attaches, struct ld_attach, entity, uint32_t, CLAPGPU_DT_U32, "attach", "entity");
    COL(s->attaches, struct ld_attach, parent, uint32_t, CLAPGPU_DT_U32, "attach", "parent");
    COL(s->attaches, struct ld_attach, joint, uint32_t, CLAPGPU_DT_U32, "attach", "joint");
    COL(s->bodies, struct ld_body, entity, uint32_t, CLAPGPU_DT_U32, "bodies", "entity");
    COL(s->bodies, struct ld_body, geom_class, int32_t, CLAPGPU_DT_I32, "bodies", "geom_class");
    COL(s->bodies, struct ld_body, phys_type, int32_t, CLAPGPU_DT_I32, "bodies", "phys_type");
    COL(s->bodies, struct ld_body, mass, double, CLAPGPU_DT_F64, "bodies", "mass");
    COL(s->bodies, struct ld_body, radius, double, CLAPGPU_DT_F64, "bodies", "radius");
    COL(s->bodies, struct ld_body, length, double, CLAPGPU_DT_F64, "bodies", "length");
    COL(s->bodies, struct ld_body, yoffset, double, CLAPGPU_DT_F64, "bodies", "yoffset");
    COL(s->bodies, struct ld_body, bounce, double, CLAPGPU_DT_F64, "bodies", "bounce");
    COL(s->bodies, struct ld_body, bounce_vel, double, CLAPGPU_DT_F64, "bodies", "bounce_vel");
    COL(s->chars, struct ld_char, entity, uint32_t, CLAPGPU_DT_U32, "characters", "entity");
    COL(s->chars, struct ld_char, model, uint32_t, CLAPGPU_DT_U32, "characters", "model");
    COL(s->chars, struct ld_char, speed, double, CLAPGPU_DT_F64, "characters", "speed");
    COL(s->chars, struct ld_char, can_jump, uint8_t, CLAPGPU_DT_U8, "characters", "can_jump");
    COL(s->chars, struct ld_char, can_dash, uint8_t, CLAPGPU_DT_U8, "characters", "can_dash");
#undef COL
#undef A
out:
    free(ps); free(rot); free(maabb); free(parent); free(pj); free(model); free(flags); free(seqs); free(mskip);
    if (rc) { clapgpu_snapshot_abort(w); return rc; }
    return clapgpu_snapshot_finish(w);
}

int clapgpu_load_scene(const char *scene_json, const char *asset_dir, const char *snapshot_path, char *err, size_t err_len)
{
    struct ld_err e = { err, err_len };
    if (err && err_len) err[0] = 0;
    if (!scene_json || !snapshot_path) return fail(&e, LD_INVALID, "missing path");
    uint8_t *buf = NULL;
    size_t size = 0;
    int rc = read_file(scene_json, &buf, &size);
    if (rc) return fail(&e, rc, "cannot read '%s'", scene_json);
    struct jparse jp;
    struct jnode *root = jdecode(&jp, (const char *)buf, size);
    if (!root) { free(buf); return fail(&e, LD_PARSE, "couldn't parse '%s'", scene_json); }
    char dir[4096];
    if (asset_dir) snprintf(dir, sizeof(dir), "%s", asset_dir);
    else {
        snprintf(dir, sizeof(dir), "%s", scene_json);
        char *slash = strrchr(dir, '/');
        if (slash) *slash = 0; else strcpy(dir, ".");
    }
    struct ld_scene s;
    memset(&s, 0, sizeof(s));
    s.models.el = sizeof(struct ld_model); s.entities.el = sizeof(struct ld_entity); s.carriers.el = sizeof(struct ld_carrier);
    s.attaches.el = sizeof(struct ld_attach); s.bodies.el = sizeof(struct ld_body); s.chars.el = sizeof(struct ld_char);
    s.asset_dir = dir;
    rc = LD_OK;
    if (root->tag != J_OBJECT) rc = fail(&e, LD_PARSE, "parse error in '%s'", scene_json);
    for (struct jnode *p = rc ? NULL : root->head; p && !rc; p = p->next) {          /* scene_onload, scene.c:1846-1873 */
        if (!strcmp(p->key, "name")) {
            if (p->tag != J_STRING) rc = fail(&e, LD_PARSE, "parse error in '%s': name", scene_json);
        } else if (!strcmp(p->key, "model")) {
            if (p->tag != J_ARRAY) { rc = fail(&e, LD_PARSE, "parse error in '%s': model", scene_json); break; }
            for (struct jnode *m = p->head; m && !rc; m = m->next) rc = model_from_json(&s, m, &e);
        } else if (!strcmp(p->key, "light") && p->tag == J_ARRAY) {
            for (struct jnode *l = p->head; l && !rc; l = l->next) rc = light_from_json(&s, l, &e);
        }
    }
    if (!rc) {
        rc = write_scene(&s, snapshot_path);
        if (rc) fail(&e, rc, "cannot write '%s'", snapshot_path);
    }
    scene_free(&s);
    jfree(&jp);
    free(buf);
    return rc;
}

int clapgpu_load_gltf(const char *gltf_path, int fix_origin, const char *snapshot_path, char *err, size_t err_len)
{
    struct ld_err e = { err, err_len };
    if (err && err_len) err[0] = 0;
    if (!gltf_path || !snapshot_path) return fail(&e, LD_INVALID, "missing path");
    struct gltf g;
    int rc = gltf_load_file(&g, gltf_path, &e);
    if (rc) return rc;
    const int mesh = gltf_pick_mesh(&g);
    if (mesh < 0) { gltf_free(&g); return fail(&e, LD_PARSE, "'%s': no mesh to instantiate", gltf_path); }
    struct ld_scene s;
    memset(&s, 0, sizeof(s));
    s.models.el = sizeof(struct ld_model); s.entities.el = sizeof(struct ld_entity); s.carriers.el = sizeof(struct ld_carrier);
    s.attaches.el = sizeof(struct ld_attach); s.bodies.el = sizeof(struct ld_body); s.chars.el = sizeof(struct ld_char);
    struct ld_model *m = vpush(&s.models);
    rc = m ? model_from_gltf(m, &g, mesh, fix_origin, &e) : LD_NOMEM;
    gltf_free(&g);
    if (!rc) {
        rc = write_scene(&s, snapshot_path);
        if (rc) fail(&e, rc, "cannot write '%s'", snapshot_path);
    }
    scene_free(&s);
    return rc;
}
