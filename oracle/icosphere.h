/* oracle/icosphere.h -- TEST INFRASTRUCTURE. Deterministic icosphere tessellation for
 * BASELINE.json configs[4] ("synthetic tessellated icosphere mesh, ~1M triangles").
 * Not in the reference: the reference only ships data/icosahedron.obj (12 v / 20 f), which
 * is the level-0 input here. Each level splits every face into 4 through edge midpoints
 * pushed onto the unit sphere; normals are the vertex positions (smooth sphere).
 * Level 8 -> 655 362 vertices, 1 310 720 faces (Config 5).
 */
#ifndef TRC_ORACLE_ICOSPHERE_H
#define TRC_ORACLE_ICOSPHERE_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct ico_mesh {
  uint32_t nverts, nfaces;
  float*    verts; /* 3 per vertex */
  uint32_t* faces; /* 3 vertex indices per face */
} ico_mesh;

static inline void ico_normalize3(float* v)
{
  volatile float s = v[0]*v[0];
  volatile float t = v[1]*v[1];
  volatile float u = v[2]*v[2];
  volatile float q = s + t;
  q = q + u;
  float inv = 1.0f / sqrtf(q);
  v[0] = v[0]*inv; v[1] = v[1]*inv; v[2] = v[2]*inv;
}

/* open-addressing edge -> midpoint-vertex map */
typedef struct ico_edge_map { uint64_t* keys; uint32_t* vals; uint64_t cap; } ico_edge_map;

static inline uint32_t ico_midpoint(ico_edge_map* m, float** verts, uint32_t* nverts, uint32_t a, uint32_t b)
{
  uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  uint64_t key = ((uint64_t)lo << 32) | hi;
  uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 20;
  for (;;) {
    h &= (m->cap - 1);
    if (m->keys[h] == key) return m->vals[h];
    if (m->keys[h] == ~0ull) break;
    ++h;
  }
  float* v = *verts;
  uint32_t id = *nverts;
  float p[3];
  for (int k = 0; k < 3; ++k) {
    volatile float s = v[3*lo+k] + v[3*hi+k];
    p[k] = s * 0.5f;
  }
  ico_normalize3(p);
  v[3*id+0] = p[0]; v[3*id+1] = p[1]; v[3*id+2] = p[2];
  *nverts = id + 1;
  m->keys[h] = key; m->vals[h] = id;
  return id;
}

/* Subdivide (verts0, faces0) `levels` times. Level-0 vertices are normalised first. */
static inline ico_mesh ico_subdivide(const float* verts0, uint32_t nverts0,
                                     const uint32_t* faces0, uint32_t nfaces0, int levels)
{
  ico_mesh cur;
  cur.nverts = nverts0; cur.nfaces = nfaces0;
  /* final counts: F = F0*4^L, V = V0 + (edges added): V_L = 10*4^L + 2 for the icosahedron */
  uint64_t maxf = (uint64_t)nfaces0 << (2*levels);
  uint64_t maxv = (uint64_t)nverts0 + 3 * maxf / 2 + 16;
  cur.verts = (float*)malloc(sizeof(float) * 3 * maxv);
  memcpy(cur.verts, verts0, sizeof(float) * 3 * nverts0);
  for (uint32_t i = 0; i < nverts0; ++i) ico_normalize3(cur.verts + 3*i);
  cur.faces = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (nfaces0 ? nfaces0 : 1));
  memcpy(cur.faces, faces0, sizeof(uint32_t) * 3 * nfaces0);

  for (int l = 0; l < levels; ++l) {
    uint64_t nedges = 3ull * cur.nfaces / 2 + 8;
    ico_edge_map m;
    m.cap = 1; while (m.cap < 2 * nedges) m.cap <<= 1;
    m.keys = (uint64_t*)malloc(sizeof(uint64_t) * m.cap);
    m.vals = (uint32_t*)malloc(sizeof(uint32_t) * m.cap);
    memset(m.keys, 0xFF, sizeof(uint64_t) * m.cap);
    uint32_t* nf = (uint32_t*)malloc(sizeof(uint32_t) * 3 * 4 * (size_t)cur.nfaces);
    for (uint32_t f = 0; f < cur.nfaces; ++f) {
      uint32_t a = cur.faces[3*f], b = cur.faces[3*f+1], c = cur.faces[3*f+2];
      uint32_t ab = ico_midpoint(&m, &cur.verts, &cur.nverts, a, b);
      uint32_t bc = ico_midpoint(&m, &cur.verts, &cur.nverts, b, c);
      uint32_t ca = ico_midpoint(&m, &cur.verts, &cur.nverts, c, a);
      uint32_t* o = nf + 12 * (size_t)f;
      o[0] = a;  o[1] = ab; o[2]  = ca;
      o[3] = b;  o[4] = bc; o[5]  = ab;
      o[6] = c;  o[7] = ca; o[8]  = bc;
      o[9] = ab; o[10] = bc; o[11] = ca;
    }
    free(cur.faces); cur.faces = nf; cur.nfaces *= 4;
    free(m.keys); free(m.vals);
  }
  return cur;
}

/* Object-space radius the scenes give the level-`levels` icosphere: 2^max(0, levels-3). Ray.inl:196
 * rejects every triangle with |det| < EPS = 0.001 (absolute) and |det| <= |e1 x e2|, so on the unit
 * sphere the faces of level 5 and above would all be rejected; the vertices are scaled by this
 * power of two (exact) and the object scaled down by it, normals stay unit vectors. */
static inline float ico_radius(int levels) { return levels > 3 ? (float)(1u << (levels - 3)) : 1.0f; }

#endif
