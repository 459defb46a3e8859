/* Design analysis (CPU): where in the pre-order node array does a lock-step group of gs key-adjacent bodies
 * spend its visits?  For every sampled group, the wave-visits per 1/NB-th of the pre-order index range
 * (NB bins), so that splits of the array among several cursors of one wave can be evaluated.
 * Build: gcc -O2 -fopenmp -shared -fPIC -o range_balance.so range_balance.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef unsigned __int128 mask_t;

/* pre-order rank of every node (children in octant order = the product's node order) */
void preorder_rank(const int32_t *children, const uint8_t *is_leaf, int64_t num_nodes, int32_t *rank) {
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * (size_t)(num_nodes + 8));
    int64_t sp = 0, next = 0;
    st[sp++] = 0;
    while (sp > 0) {
        int32_t node = st[--sp];
        rank[node] = (int32_t)next++;
        if (is_leaf[node]) continue;
        for (int c = 7; c >= 0; c--) { int32_t ch = children[8 * (int64_t)node + c]; if (ch >= 0) st[sp++] = ch; }
    }
    free(st);
}

void range_visits(const double *pos, const int64_t *order, int64_t n, int gs, const double *half, const double *com,
                  const int32_t *children, const uint8_t *is_leaf, const int32_t *rank, int64_t num_nodes, double theta,
                  double softening, int nb, int64_t *out /* groups x nb */) {
    const double eps2 = softening * softening;
    int64_t ngroups = n / gs;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t g = 0; g < ngroups; g++) {
        int64_t lo = g * gs;
        int cap = 8192, sp = 0;
        int32_t *sn = (int32_t *)malloc(sizeof(int32_t) * cap);
        mask_t *sm = (mask_t *)malloc(sizeof(mask_t) * cap);
        sn[0] = 0; sm[0] = (gs == 128) ? ~(mask_t)0 : ((((mask_t)1) << gs) - 1); sp = 1;
        int64_t *o = out + g * nb;
        for (int b = 0; b < nb; b++) o[b] = 0;
        while (sp > 0) {
            sp--;
            int32_t node = sn[sp];
            mask_t mask = sm[sp], open = 0;
            o[(int64_t)rank[node] * nb / num_nodes]++;
            if (is_leaf[node]) continue;
            for (int l = 0; l < gs; l++) {
                if (!((mask >> l) & 1)) continue;
                int64_t i = order[lo + l];
                double dx = com[3 * node] - pos[3 * i], dy = com[3 * node + 1] - pos[3 * i + 1], dz = com[3 * node + 2] - pos[3 * i + 2];
                double dist = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                if (!(half[node] * 2.0 / dist < theta)) open |= ((mask_t)1) << l;
            }
            if (!open) continue;
            for (int c = 0; c < 8; c++) {
                int32_t ch = children[8 * (int64_t)node + c];
                if (ch < 0) continue;
                if (sp >= cap) { cap *= 2; sn = (int32_t *)realloc(sn, sizeof(int32_t) * cap); sm = (mask_t *)realloc(sm, sizeof(mask_t) * cap); }
                sn[sp] = ch; sm[sp] = open; sp++;
            }
        }
        free(sn); free(sm);
    }
}

/* every visit's pre-order rank, per group (at most cap per group; counts[g] = number of visits) */
void visit_ranks(const double *pos, const int64_t *order, int64_t n, int gs, const double *half, const double *com,
                 const int32_t *children, const uint8_t *is_leaf, const int32_t *rank, double theta, double softening,
                 int64_t cap_out, int32_t *out /* groups x cap_out */, int64_t *counts) {
    const double eps2 = softening * softening;
    int64_t ngroups = n / gs;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t g = 0; g < ngroups; g++) {
        int64_t lo = g * gs;
        int cap = 8192, sp = 0;
        int32_t *sn = (int32_t *)malloc(sizeof(int32_t) * cap);
        mask_t *sm = (mask_t *)malloc(sizeof(mask_t) * cap);
        sn[0] = 0; sm[0] = (gs == 128) ? ~(mask_t)0 : ((((mask_t)1) << gs) - 1); sp = 1;
        int64_t k = 0;
        while (sp > 0) {
            sp--;
            int32_t node = sn[sp];
            mask_t mask = sm[sp], open = 0;
            if (k < cap_out) out[g * cap_out + k] = rank[node];
            k++;
            if (is_leaf[node]) continue;
            for (int l = 0; l < gs; l++) {
                if (!((mask >> l) & 1)) continue;
                int64_t i = order[lo + l];
                double dx = com[3 * node] - pos[3 * i], dy = com[3 * node + 1] - pos[3 * i + 1], dz = com[3 * node + 2] - pos[3 * i + 2];
                double dist = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                if (!(half[node] * 2.0 / dist < theta)) open |= ((mask_t)1) << l;
            }
            if (!open) continue;
            for (int c = 0; c < 8; c++) {
                int32_t ch = children[8 * (int64_t)node + c];
                if (ch < 0) continue;
                if (sp >= cap) { cap *= 2; sn = (int32_t *)realloc(sn, sizeof(int32_t) * cap); sm = (mask_t *)realloc(sm, sizeof(mask_t) * cap); }
                sn[sp] = ch; sm[sp] = open; sp++;
            }
        }
        counts[g] = k;
        free(sn); free(sm);
    }
}

/* the leaf node of every body */
void body_leaves(const int32_t *children, const uint8_t *is_leaf, const int32_t *leaf_body, int64_t num_nodes, int32_t *leaf_of_body) {
    for (int64_t i = 0; i < num_nodes; i++) if (is_leaf[i] && leaf_body[i] >= 0) leaf_of_body[leaf_body[i]] = (int32_t)i;
}
