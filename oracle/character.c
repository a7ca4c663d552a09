/*
 * oracle/character.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * The per-character feeder that runs in front of default_update: character_update
 * (character.c:583-611) with its position history (history_push / _fetch / _newest,
 * character.c:546-581, POS_HISTORY_MAX = 8, character.h:21) and the body read-back it does itself
 * (phys_body_update / phys_body_set_position, physics.c:789-812, 208-225).  character_motion_reset
 * only acts on the one controlled character and stays host code.
 *
 * Pinned against the reference for characters without a body (oracle/ref harness "characters" runs
 * the real character_update -> default_update); the body branch calls into ODE, so for it only the
 * reference's own arithmetic around the ODE getters/setters is restated (PARITY UNPINNED there).
 */
#include "clap_oracle.h"
#include "lm.h"

#define HIST 8

/*
 * char_entity[c]: entity of character c; char_body[c]: its body or -1 (no ENTITY3D_HAS_PHYSICS).
 * hist_pos[c][8][3], hist_head[c], hist_wrapped[c]: character.history; airborne[c]: character.airborne.
 * moved[c] (out): phys_body_update() returned true (character_set_moved: the host marks the
 * character's camera updated).  Characters are visited in list order.
 */
void clapo_characters_update(uint32_t n_chars, const uint32_t *char_entity, const int32_t *char_body,
                             float limbo_height, float *hist_pos, uint32_t *hist_head, uint8_t *hist_wrapped,
                             const uint8_t *airborne, float *pos_scale, uint32_t *entity_flags,
                             double *body_pos, const double *body_lvel, const double *body_yoffset,
                             uint8_t *moved)
{
    for (uint32_t c = 0; c < n_chars; c++) {
        const uint32_t e = char_entity[c];
        const int32_t b = char_body[c];
        float *hp = hist_pos + (size_t)c * HIST * 3;
        float *pos = pos_scale + 4 * (size_t)e;
        float last[3] = { 0.f, 0.f, 0.f };

        /* history_newest */
        if (hist_head[c]) memcpy(last, hp + 3 * (hist_head[c] - 1), 12);
        else if (hist_wrapped[c]) memcpy(last, hp + 3 * (HIST - 1), 12);

        /* fell too far below the last grounded position: teleport back (character.c:595-599) */
        if (lm_dot3(last, last) > 0.0 && fabsf(pos[1] - last[1]) >= limbo_height) {
            float p[3];
            if (hist_wrapped[c]) {                                  /* history_fetch */
                memcpy(p, hp + 3 * hist_head[c], 12);
                hist_wrapped[c] = 0;
            } else {
                memcpy(p, hp, 12);
            }
            hist_head[c] = 0;
            pos[0] = p[0]; pos[1] = p[1]; pos[2] = p[2];            /* entity3d_position: transform_set_pos ... */
            entity_flags[e] |= CLAPO_E_DIRTY;
            if (b >= 0) {                                           /* ... + phys_body_set_position */
                body_pos[3 * (size_t)b + 0] = p[0];
                body_pos[3 * (size_t)b + 1] = p[1] + body_yoffset[b];
                body_pos[3 * (size_t)b + 2] = p[2];
            }
        }

        moved[c] = 0;
        if (b >= 0) {                                               /* phys_body_update (characters keep their rotation) */
            const double *bp = body_pos + 3 * (size_t)b, *v = body_lvel + 3 * (size_t)b;
            pos[0] = bp[0];
            pos[1] = bp[1] - body_yoffset[b];
            pos[2] = bp[2];
            entity_flags[e] |= CLAPO_E_DIRTY;
            if (sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) > 1e-3) {
                if (!airborne[c]) {                                 /* history_push */
                    memcpy(hp + 3 * hist_head[c], pos, 12);
                    hist_head[c] = (hist_head[c] + 1) % HIST;
                    if (!hist_wrapped[c] && !hist_head[c]) hist_wrapped[c] = 1;
                }
                moved[c] = 1;
            }
        }
    }
}
