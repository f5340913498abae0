/*
 * mc_oracle.c -- CPU oracle for extract_surface (marching cubes), SURVEY.md section 8 row f2.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see tsdf_oracle.h).  A plain-C restatement of the reference's
 * src/MarchingCubes/MarkAndSweepMC.cu, stage by stage:
 *   voxel_indices_for_cube_index   :65-99    cube index -> its 8 voxels, corner numbering of the diagram (:9-36)
 *   calculate_cube_type            :110-124  bit i = (distance of corner i < 0)
 *   get_cube_contribution          :133-152  vertices per cube = VERTICES_FOR_CUBE_TYPE[type] (MC_triangle_table.cu:46)
 *   extract_surface_ms             :390-500  host scan of the occupied cubes -> write offsets (:456-474)
 *   interpolate                    :47-63    zero crossing on an edge (swap rule, ratio = -w0 / (w1 - w0))
 *   generate_vertices              :219-312  8 voxel centres (centre_of_voxel_at, src/TSDF/TSDF_utilities.cu:10-17), 12 edge
 *                                            points in the EDGE_VERTICES order (:291-302, MC_edge_table.cu:47), emitted by walking
 *                                            TRIANGLE_TABLE[type] until -1 (:285)
 *   extract_surface                :506-555  vertices in that order; triangle t = (3t, 3t+2, 3t+1) (:549)
 *
 * Tables.  The reference's TRIANGLE_TABLE is P. Bourke's published table ("Polygonising a scalar field", 1994).  It is
 * rebuilt here from 30 base configurations carried round the cube by its 24 rotations (the construction of the classic
 * tables) and PINNED on the reference's own file: tests/test_oracle_pins.py compares the SHA-256 of the 256 x 16 table
 * and of the derived vertex counts with tests/golden/mc_tables.sha256.json, which tools/mc_table_sha.py computes from
 * /root/reference/src/MarchingCubes/MC_triangle_table.cu where it lies.  The arithmetic (interpolate, voxel centres)
 * has no reference-produced vectors: like integrate / raycast it is a line-by-line restatement (parity unpinned beyond
 * the tables and the closed-form checks of tests/test_oracle_pins.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tsdf_oracle.h"

/* corner i of the cube rooted at voxel (x, y, z) sits at voxel (x + dx, y + dy, z + dz)   (MarkAndSweepMC.cu:80-97) */
static const int CORNER[8][3] = {{0, 0, 1}, {1, 0, 1}, {1, 0, 0}, {0, 0, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}, {0, 1, 0}};
/* EDGE_VERTICES (MC_edge_table.cu:47): the two corners of each edge, in the order generate_vertices interpolates them */
static const int EDGE_VERTICES[12][2] = {{0, 1}, {2, 1}, {3, 2}, {3, 0}, {4, 5}, {6, 5}, {7, 6}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

/* ---- table construction ------------------------------------------------------------------------------------------
 * A rotation of the cube is a signed permutation of the axes with determinant +1.  TRY_ORDER lists the 24 of them as
 * {axis that feeds x, y, z} and a sign mask (bit k: output axis k is mirrored); a configuration is rotated by each in turn
 * until it equals a base configuration, whose triangles are then rotated back. */
static const struct { uint8_t src[3]; uint8_t flip; } TRY_ORDER[24] = {
    {{0, 1, 2}, 0}, {{1, 2, 0}, 3}, {{2, 0, 1}, 6}, {{0, 2, 1}, 7}, {{1, 0, 2}, 2}, {{0, 1, 2}, 5}, {{2, 0, 1}, 3}, {{2, 1, 0}, 1},
    {{1, 2, 0}, 6}, {{2, 1, 0}, 4}, {{0, 2, 1}, 2}, {{1, 0, 2}, 7}, {{2, 0, 1}, 5}, {{0, 1, 2}, 3}, {{1, 2, 0}, 0}, {{1, 0, 2}, 1},
    {{0, 2, 1}, 4}, {{1, 0, 2}, 4}, {{0, 2, 1}, 1}, {{2, 1, 0}, 2}, {{2, 1, 0}, 7}, {{2, 0, 1}, 0}, {{0, 1, 2}, 6}, {{1, 2, 0}, 5}};

/* base configurations: bit i = corner i negative; one hex digit per triangle corner = the edge it lies on */
static const struct { uint8_t config; const char *triangles; } BASE[30] = {
    {0, ""}, {1, "083"}, {3, "183981"}, {5, "08312a"}, {7, "2832a8a98"},
    {15, "98aa8b"}, {20, "12a847"}, {21, "34730412a"}, {23, "2a9297273794"}, {27, "47b94b9b2921"},
    {31, "47b4b99ba"}, {37, "30812a495"}, {45, "4950818a18ba"}, {60, "958857a13a3b"}, {61, "5705097b010aba0"},
    {63, "ba57b5"}, {90, "01947823b5a6"}, {92, "8473b53515b6"}, {94, "059065036b63847"}, {95, "65969b4797b9"},
    {113, "0730a709a67a"}, {125, "091b67"}, {141, "a7617a187108"}, {142, "03707a0a96a7"}, {150, "4b846b0292a9"},
    {153, "042462"}, {165, "6b712a083495"}, {191, "a56"}, {232, "29a279237749"}, {255, ""}};

static int corner_at(int x, int y, int z) {
    for (int i = 0; i < 8; i++)
        if (CORNER[i][0] == x && CORNER[i][1] == y && CORNER[i][2] == z) return i;
    return -1;
}

static int edge_of(int a, int b) {
    for (int e = 0; e < 12; e++)
        if ((EDGE_VERTICES[e][0] == a && EDGE_VERTICES[e][1] == b) || (EDGE_VERTICES[e][0] == b && EDGE_VERTICES[e][1] == a)) return e;
    return -1;
}

/* where corner i goes under rotation r */
static int rotate_corner(int r, int i) {
    int p[3];
    for (int k = 0; k < 3; k++) {
        const int v = CORNER[i][TRY_ORDER[r].src[k]];
        p[k] = ((TRY_ORDER[r].flip >> k) & 1) ? 1 - v : v;
    }
    return corner_at(p[0], p[1], p[2]);
}

void orc_mc_tables(int8_t triangle_table[256][16], uint8_t vertices_for_cube_type[256]) {
    int base_of[256];
    for (int c = 0; c < 256; c++) base_of[c] = -1;
    for (int b = 0; b < 30; b++) base_of[BASE[b].config] = b;
    for (int c = 0; c < 256; c++) {
        int n = 0;
        for (int r = 0; r < 24; r++) {
            int image[8], back[8], turned = 0;
            for (int i = 0; i < 8; i++) {
                image[i] = rotate_corner(r, i);
                back[image[i]] = i;
                if ((c >> i) & 1) turned |= 1 << image[i];
            }
            if (base_of[turned] < 0) continue;
            for (const char *d = BASE[base_of[turned]].triangles; *d; d++) {
                const int e = (*d <= '9') ? *d - '0' : *d - 'a' + 10;
                triangle_table[c][n++] = (int8_t)edge_of(back[EDGE_VERTICES[e][0]], back[EDGE_VERTICES[e][1]]);
            }
            break;
        }
        vertices_for_cube_type[c] = (uint8_t)n;
        for (; n < 16; n++) triangle_table[c][n] = -1;
    }
}

/* ---- the reference's stages ----------------------------------------------------------------------------------- */

/* voxel_indices_for_cube_index (:65-99) */
static void voxel_indices_for_cube_index(int64_t cube_index, int gx, int gy, int64_t voxel_indices[8], unsigned voxel_coords[8][3]) {
    const int64_t slab = (int64_t)(gx - 1) * (gy - 1);
    const unsigned cz = (unsigned)(cube_index / slab);
    const unsigned cy = (unsigned)((cube_index - (int64_t)cz * slab) / (gx - 1));
    const unsigned cx = (unsigned)((cube_index - (int64_t)cz * slab) % (gx - 1));
    const int64_t dx = 1, dy = gx, dz = dy * gy;
    const int64_t root = (int64_t)cz * gx * gy + (int64_t)cy * gx + cx;   /* front left bottom */
    voxel_indices[0] = root + dz;
    voxel_indices[1] = root + dx + dz;
    voxel_indices[2] = root + dx;
    voxel_indices[3] = root;
    voxel_indices[4] = voxel_indices[0] + dy;
    voxel_indices[5] = voxel_indices[1] + dy;
    voxel_indices[6] = voxel_indices[2] + dy;
    voxel_indices[7] = voxel_indices[3] + dy;
    if (voxel_coords) {
        const unsigned c[8][3] = {{cx, cy, cz + 1}, {cx + 1, cy, cz + 1}, {cx + 1, cy, cz}, {cx, cy, cz},
                                  {cx, cy + 1, cz + 1}, {cx + 1, cy + 1, cz + 1}, {cx + 1, cy + 1, cz}, {cx, cy + 1, cz}};
        memcpy(voxel_coords, c, sizeof(c));
    }
}

/* calculate_cube_type (:110-124) */
static int calculate_cube_type(const int64_t vi[8], const float *d) {
    int t = 0;
    for (int i = 0; i < 8; i++) t |= (d[vi[i]] < 0) << i;
    return t;
}

typedef struct { float x, y, z; } f3;

/* interpolate (:47-63) with f3_sub / f3_mul_scalar / f3_add of cuda_utilities.hpp:34-59 (f3_add returns f2 + f1) */
static f3 interpolate(f3 v0, f3 v1, float w0, float w1) {
    if ((w0 > 0) && (w1 < 0)) {
        float tw = w0; w0 = w1; w1 = tw;
        f3 tv = v0; v0 = v1; v1 = tv;
    }
    const f3 delta = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z};
    const float ratio = -(w0) / (w1 - w0);
    const f3 scaled = {delta.x * ratio, delta.y * ratio, delta.z * ratio};
    const f3 r = {scaled.x + v0.x, scaled.y + v0.y, scaled.z + v0.z};
    return r;
}

/* centre_of_voxel_at (src/TSDF/TSDF_utilities.cu:10-17) */
static f3 centre_of_voxel_at(int x, int y, int z, const float vs[3], const float off[3]) {
    const f3 c = {(x + 0.5f) * vs[0] + off[0], (y + 0.5f) * vs[1] + off[1], (z + 0.5f) * vs[2] + off[2]};
    return c;
}

/* extract_surface (:506-555) = extract_surface_ms (:390-500) + the copy to the host.  vertices == NULL: count only.
 * Returns the number of vertices (three per triangle), or -1 when `capacity` is too small. */
int64_t orc_marching_cubes(const float *dist, uint32_t X, uint32_t Y, uint32_t Z, const float vs[3], const float offset[3],
                           float *vertices, int64_t capacity, int nthreads) {
    if (X < 2 || Y < 2 || Z < 2) return 0;
    static int8_t TRIANGLE_TABLE[256][16];
    static uint8_t VERTICES_FOR_CUBE_TYPE[256];
    static int have_tables = 0;
    if (!have_tables) {
        orc_mc_tables(TRIANGLE_TABLE, VERTICES_FOR_CUBE_TYPE);
        have_tables = 1;
    }
    const int64_t max_cubes = (int64_t)(X - 1) * (Y - 1) * (Z - 1);
    if (nthreads < 1) nthreads = 1;

    /* get_cube_contribution (:133-152) for every cube */
    uint8_t *vertices_per_cube = (uint8_t *)malloc((size_t)max_cubes);
    if (!vertices_per_cube) return -2;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t cube = 0; cube < max_cubes; cube++) {
        int64_t vi[8];
        voxel_indices_for_cube_index(cube, (int)X, (int)Y, vi, NULL);
        vertices_per_cube[cube] = VERTICES_FOR_CUBE_TYPE[calculate_cube_type(vi, dist)];
    }
    /* launch_get_cube_contribution's host count (:196-204) and the scan of extract_surface_ms (:456-474) */
    int64_t num_occupied = 0, num_vertices = 0;
    for (int64_t i = 0; i < max_cubes; i++) {
        num_vertices += vertices_per_cube[i];
        if (vertices_per_cube[i] > 0) num_occupied++;
    }
    if (!vertices) {
        free(vertices_per_cube);
        return num_vertices;
    }
    if (num_vertices > capacity) {
        free(vertices_per_cube);
        return -1;
    }
    int64_t *thread_write_offsets = (int64_t *)malloc(sizeof(int64_t) * (size_t)(num_occupied ? num_occupied : 1));
    int64_t *cube_indices = (int64_t *)malloc(sizeof(int64_t) * (size_t)(num_occupied ? num_occupied : 1));
    int64_t current_offset = 0, output_index = 0;
    for (int64_t cube = 0; cube < max_cubes; cube++)
        if (vertices_per_cube[cube] > 0) {
            thread_write_offsets[output_index] = current_offset;
            cube_indices[output_index] = cube;
            current_offset += vertices_per_cube[cube];
            output_index++;
        }
    free(vertices_per_cube);

    /* generate_vertices (:219-312), one "thread" per occupied cube */
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t data_index = 0; data_index < num_occupied; data_index++) {
        int64_t vi[8];
        unsigned vc[8][3];
        voxel_indices_for_cube_index(cube_indices[data_index], (int)X, (int)Y, vi, vc);
        float w[8];
        f3 v[8];
        for (int i = 0; i < 8; i++) {
            w[i] = dist[vi[i]];
            v[i] = centre_of_voxel_at((int)vc[i][0], (int)vc[i][1], (int)vc[i][2], vs, offset);
        }
        f3 vertex[12];
        for (int e = 0; e < 12; e++) {
            const int a = EDGE_VERTICES[e][0], b = EDGE_VERTICES[e][1];
            vertex[e] = interpolate(v[a], v[b], w[a], w[b]);
        }
        int64_t out = thread_write_offsets[data_index];
        const int cube_type = calculate_cube_type(vi, dist);
        int edge_index, i = 0;
        while ((edge_index = TRIANGLE_TABLE[cube_type][i]) != -1) {
            vertices[3 * out + 0] = vertex[edge_index].x;
            vertices[3 * out + 1] = vertex[edge_index].y;
            vertices[3 * out + 2] = vertex[edge_index].z;
            out++;
            i++;
        }
    }
    free(thread_write_offsets);
    free(cube_indices);
    return num_vertices;
}
