// skin.hip -- vertex skinning for gfx950.
//
// Materialises what the reference leaves to its vertex shader (shaders/model.vert:32-48, also
// shadow.vert:19-23 and shadow_vsm.vert:19-23, i.e. recomputed in every geometry pass):
//     p' = sum_{i<4} w_i * (J[j_i] * (p,1)),   n' = sum_{i<4} w_i * (J[j_i] * (n,0))
// with J = the entity's joint_transforms, accumulated i = 0..3 in order, fp32, no weight
// renormalisation.  The shader goes on with the vec4 (model.vert:44: proj * view * trs * total_local_pos), whose
// w = sum_i w_i * (row 3 of J[j_i] . (p,1)) -- the sum of the weights for affine palettes, NOT 1 unless the
// asset's weights are normalised: out_w (optional) carries it, so a pre-skinned draw can feed the same vec4.
// Inputs are the reference's vertex attribute formats (mesh.h:125-131,
// gltf.c:387-388): position f32x3, normal f32x3, joints u8x4, weights f32x4.
//
// Mapping: one workgroup per character; its palette (nr_joints x 64 B) is staged once in LDS
// with an 80-byte row pitch (conflict-free 16-B column reads for lanes hitting different
// joints) and every lane skins one vertex.  HBM: 44 B in + 24 B out per vertex + the palette
// once per character (88.5 B / vertex at 64 joints, 200 vertices).
#include "common.h"
#include "lm_dev.h"

namespace clapgpu {

constexpr int SKIN_BLOCK = 256;
constexpr int PAL_PITCH = 20;          // floats per palette row in LDS (16 + 4 pad)
constexpr int PAL_MAX_JOINTS = 256;

struct SkinArgs {
    uint32_t        n_chars, J;
    const uint32_t *vert_first, *vert_count, *out_first;
    const float    *position, *normal;
    const uint32_t *joints;            // u8x4 packed
    const float4   *weights;
    const float4   *joint_transforms;
    float          *out_position, *out_normal, *out_w;
};

struct SkinVert {
    float    px, py, pz, nx, ny, nz;
    uint32_t jj;
    float4   w;
};

__device__ __forceinline__ SkinVert load_vert(const SkinArgs &a, size_t v)
{
    // vertex attributes are read once per frame: non-temporal loads keep them from evicting the
    // palette (and everything else that is reused) from the infinity cache
    SkinVert r;
    r.px = __builtin_nontemporal_load(&a.position[3 * v]);
    r.py = __builtin_nontemporal_load(&a.position[3 * v + 1]);
    r.pz = __builtin_nontemporal_load(&a.position[3 * v + 2]);
    r.nx = __builtin_nontemporal_load(&a.normal[3 * v]);
    r.ny = __builtin_nontemporal_load(&a.normal[3 * v + 1]);
    r.nz = __builtin_nontemporal_load(&a.normal[3 * v + 2]);
    r.jj = __builtin_nontemporal_load(&a.joints[v]);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 w = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&a.weights[v]));
    r.w = make_float4(w.x, w.y, w.z, w.w);
    return r;
}

constexpr int SKIN_WAVES = 8;           // 56 VGPRs: eight workgroups per CU keep the vertex stream in flight
template <bool W>                      // W: also total_local_pos.w (4 B / vertex more)
__global__ __launch_bounds__(SKIN_BLOCK, SKIN_WAVES)
void k_skin(SkinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float pal[];   // J rows of PAL_PITCH floats (5 KiB at 64 joints)

    const uint32_t c = blockIdx.x;
    const uint32_t J = a.J;
    const uint32_t vfirst = a.vert_first[c], vcount = a.vert_count[c], ofirst = a.out_first[c];

    // stage the palette: J * 4 float4, coalesced
    const float4 *src = a.joint_transforms + (size_t)c * J * 4;
    for (uint32_t q = threadIdx.x; q < J * 4; q += SKIN_BLOCK) {
        const float4 v = src[q];
        *reinterpret_cast<float4 *>(pal + (q >> 2) * PAL_PITCH + (q & 3) * 4) = v;
    }
    uint32_t k = threadIdx.x;
    SkinVert cur;
    if (k < vcount)
        cur = load_vert(a, (size_t)vfirst + k);                   // in flight across the barrier, beside the palette loads
    __syncthreads();

    while (k < vcount) {
        const float w[4] = { cur.w.x, cur.w.y, cur.w.z, cur.w.w };
        float tp[3] = { 0, 0, 0 }, tn[3] = { 0, 0, 0 }, tw = 0.f;
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
            const uint32_t ji = (cur.jj >> (8 * i)) & 0xffu;
            const float4 *m = reinterpret_cast<const float4 *>(pal + ji * PAL_PITCH);
            const float4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
            const float wi = i == 0 ? w[0] : i == 1 ? w[1] : i == 2 ? w[2] : w[3];
            // ((M0 x + M1 y) + M2 z) + M3 w per component; w = 1 for positions, 0 for normals
            const float lx = ((c0.x * cur.px + c1.x * cur.py) + c2.x * cur.pz) + c3.x * 1.0f;
            const float ly = ((c0.y * cur.px + c1.y * cur.py) + c2.y * cur.pz) + c3.y * 1.0f;
            const float lz = ((c0.z * cur.px + c1.z * cur.py) + c2.z * cur.pz) + c3.z * 1.0f;
            const float mx = ((c0.x * cur.nx + c1.x * cur.ny) + c2.x * cur.nz) + c3.x * 0.0f;
            const float my = ((c0.y * cur.nx + c1.y * cur.ny) + c2.y * cur.nz) + c3.y * 0.0f;
            const float mz = ((c0.z * cur.nx + c1.z * cur.ny) + c2.z * cur.nz) + c3.z * 0.0f;
            tp[0] += lx * wi; tp[1] += ly * wi; tp[2] += lz * wi;
            tn[0] += mx * wi; tn[1] += my * wi; tn[2] += mz * wi;
            if (W) {
                const float lw = ((c0.w * cur.px + c1.w * cur.py) + c2.w * cur.pz) + c3.w * 1.0f;
                tw += lw * wi;
            }
        }
        const size_t o = (size_t)ofirst + k;
        // written once, read by the draw path: streaming stores keep them out of the infinity cache
        __builtin_nontemporal_store(tp[0], &a.out_position[3 * o]);
        __builtin_nontemporal_store(tp[1], &a.out_position[3 * o + 1]);
        __builtin_nontemporal_store(tp[2], &a.out_position[3 * o + 2]);
        __builtin_nontemporal_store(tn[0], &a.out_normal[3 * o]);
        __builtin_nontemporal_store(tn[1], &a.out_normal[3 * o + 1]);
        __builtin_nontemporal_store(tn[2], &a.out_normal[3 * o + 2]);
        if (W)
            __builtin_nontemporal_store(tw, &a.out_w[o]);
        k += SKIN_BLOCK;
        if (k < vcount)
            cur = load_vert(a, (size_t)vfirst + k);
    }
}

} // namespace clapgpu

using namespace clapgpu;

extern "C" int clapgpu_skin(void *stream, const clapgpu_skin_batch *b)
{
    if (!b)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->n_chars == 0)
        return CLAPGPU_OK;
    if (!b->vert_first || !b->vert_count || !b->out_first || !b->position || !b->normal || !b->joints ||
        !b->weights || !b->joint_transforms || !b->out_position || !b->out_normal)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->nr_joints == 0 || b->nr_joints > PAL_MAX_JOINTS)
        return CLAPGPU_ERR_TOO_LARGE;
    SkinArgs a;
    a.n_chars = b->n_chars;
    a.J = b->nr_joints;
    a.vert_first = b->vert_first;
    a.vert_count = b->vert_count;
    a.out_first = b->out_first;
    a.position = b->position;
    a.normal = b->normal;
    a.joints = reinterpret_cast<const uint32_t *>(b->joints);
    a.weights = reinterpret_cast<const float4 *>(b->weights);
    a.joint_transforms = reinterpret_cast<const float4 *>(b->joint_transforms);
    a.out_position = b->out_position;
    a.out_normal = b->out_normal;
    a.out_w = b->out_w;
    if (b->out_w)
        hipLaunchKernelGGL(k_skin<true>, dim3(b->n_chars), dim3(SKIN_BLOCK), b->nr_joints * PAL_PITCH * sizeof(float),
                           as_stream(stream), a);
    else
        hipLaunchKernelGGL(k_skin<false>, dim3(b->n_chars), dim3(SKIN_BLOCK), b->nr_joints * PAL_PITCH * sizeof(float),
                           as_stream(stream), a);
    CLAPGPU_LAUNCH_CHECK("k_skin");
    return CLAPGPU_OK;
}
