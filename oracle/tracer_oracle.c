/* oracle/tracer_oracle.c -- TEST INFRASTRUCTURE (never shipped, never on the product path).
 *
 * Plain-C, GLM-free CPU restatement of the reference's per-bounce path-tracing loop
 * (nopjia/tracer). Every function cites the reference lines it follows; paths are relative
 * to /root/reference. Vendored GLM 0.9.4.3 operation ORDER is reproduced exactly, because
 * primary-hit parity is bit-level: build with -O2 -ffp-contract=off.
 *
 * Pinning: the reference has no tests or golden vectors. This file is pinned
 *   (1) against oracle/_ref/libtracer_ref.so -- the reference's own .inl/.cpp + GLM compiled
 *       with g++ in this container -- bit for bit (tests/test_oracle_vs_ref.py), and
 *   (2) against fixtures that build generated (tests/golden/, tools/make_golden.py).
 * The RNG stream is injected (oracle/rng.h): cuRAND is outside /root/reference.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle_api.h"
#include "rng.h"
#include "icosphere.h"
#include "pmath.h"

/* ---- constants: src/common.h:17,31,40,41 ------------------------------------------- */
#define R_M_PI      3.14159265358979323846264338327950288f
#define R_EPS       0.001f
#define R_FOV       30
#define R_PATH_DEPTH 5

typedef unsigned int uint;

/* ---- types in the reference's host memory layout (LP64) ------------------------------ */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { float x, y, z, w; } v4;
typedef struct { v4 c[4]; } m4;                      /* glm::mat4: c[col], column-major */

enum { MESH = 0, SPHERE = 1, CUBE = 2, PLANE = 3 };  /* Mesh.inl:32 */
enum { DIFF = 0, TRANS = 1, MIRR = 2 };              /* Material.inl:19 */

typedef struct { uint v[3]; uint n[3]; } Face;       /* Mesh.inl:22-25 */
typedef struct { v3 v[3]; v3 n[3]; } Triangle;       /* Mesh.inl:27-30 */
typedef struct {                                     /* Mesh.inl:34-41 */
  uint numVerts, numNorms, numFaces;
  v3* verts; v3* norms; Face* faces;
  v3 bmin, bmax;
  int type;
} Mesh;
typedef struct { v3 color; float emit, brdf, n; int type; } Material; /* Material.inl:21-34 */
typedef struct { m4 matrix; m4 matrixi; Mesh* mesh; Material material; } Object; /* Object.inl:24-29 */

_Static_assert(sizeof(Face) == 24 && sizeof(Triangle) == 72, "layout");
_Static_assert(sizeof(Mesh) == 72 && sizeof(Material) == 28 && sizeof(Object) == 168, "layout");

typedef struct { v3 pos, dir; } Ray;                              /* Ray.inl:24-27 */
typedef struct { float t; int id; v3 pos, nor; } Hit;             /* Ray.inl:29-36 */

static int g_math_mode = ORC_MATH_LIBM;

/* ---- libm seam: glm::sin/cos/acos/pow -> std:: (func_trigonometric.inl:70,87,134; func_exponential.inl:41) */
static inline float m_sin(float x) { if (g_math_mode) { float s, c; pm_sincosf(x, &s, &c); return s; } return sinf(x); }
static inline float m_cos(float x) { if (g_math_mode) { float s, c; pm_sincosf(x, &s, &c); return c; } return cosf(x); }
static inline float m_acos(float x) { return g_math_mode ? pm_acosf(x) : acosf(x); }
static inline float m_pow(float x, float y) { return g_math_mode ? pm_powf(x, y) : powf(x, y); }

/* ---- GLM vector algebra, exact operation order --------------------------------------- */
static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3s(float s) { return V3(s, s, s); }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 div3(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 scl3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float gmin(float x, float y) { return x < y ? x : y; }   /* func_common.inl:296 */
static inline float gmax(float x, float y) { return x > y ? x : y; }   /* func_common.inl:314 */
static inline float gclamp(float x, float lo, float hi) { return gmin(hi, gmax(lo, x)); } /* :335 */
static inline float gabs(float x) { return x >= 0.0f ? x : -x; }       /* func_common.inl:44 */
static inline v3 min3(v3 a, v3 b) { return V3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
static inline v3 max3(v3 a, v3 b) { return V3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
/* func_geometric.inl:163 */
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* func_geometric.inl:204-207 */
static inline v3 cross3(v3 x, v3 y) {
  return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* func_exponential.inl:151: 1 / sqrt(x) */
static inline float inversesqrt(float x) { return 1.0f / sqrtf(x); }
/* func_geometric.inl:243-244 */
static inline v3 normalize3(v3 x) { float sqr = x.x * x.x + x.y * x.y + x.z * x.z; return scl3(x, inversesqrt(sqr)); }
/* func_geometric.inl:279: I - N * dot(N, I) * genType(2) */
static inline v3 reflect3(v3 I, v3 N) { return sub3(I, mul3(scl3(N, dot3(N, I)), v3s(2.0f))); }
/* func_geometric.inl:311-317 */
static inline v3 refract3(v3 I, v3 N, float eta) {
  float dotValue = dot3(N, I);
  float k = 1.0f - eta * eta * (1.0f - dotValue * dotValue);
  if (k < 0.0f) return v3s(0.0f);
  return sub3(scl3(I, eta), scl3(N, eta * dotValue + sqrtf(k)));
}
/* func_trigonometric.inl:41,56 */
static inline float radiansf(float deg) { const float pi = 3.1415926535897932384626433832795f; return deg * (pi / 180.0f); }
static inline float degreesf(float rad) { const float pi = 3.1415926535897932384626433832795f; return rad * (180.0f / pi); }

static inline v4 V4(float x, float y, float z, float w) { v4 r = { x, y, z, w }; return r; }
static inline v4 add4(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 sub4(v4 a, v4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline v4 mul4(v4 a, v4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 scl4(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline float dot4(v4 a, v4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline float* mcol(m4* m, int c) { return &m->c[c].x; }
#define M(m, col, row) ((&(m).c[col].x)[row])

static m4 m4_identity(void) {
  m4 r; memset(&r, 0, sizeof r);
  r.c[0].x = 1.0f; r.c[1].y = 1.0f; r.c[2].z = 1.0f; r.c[3].w = 1.0f; return r;
}
/* type_mat4x4.inl:689-700 (m * vec4) */
static inline v4 m4_mulv(const m4* m, v4 v) {
  return V4(
    M(*m,0,0) * v.x + M(*m,1,0) * v.y + M(*m,2,0) * v.z + M(*m,3,0) * v.w,
    M(*m,0,1) * v.x + M(*m,1,1) * v.y + M(*m,2,1) * v.z + M(*m,3,1) * v.w,
    M(*m,0,2) * v.x + M(*m,1,2) * v.y + M(*m,2,2) * v.z + M(*m,3,2) * v.w,
    M(*m,0,3) * v.x + M(*m,1,3) * v.y + M(*m,2,3) * v.z + M(*m,3,3) * v.w);
}
/* type_mat4x4.inl:757-778 (m1 * m2) */
static m4 m4_mul(const m4* m1, const m4* m2) {
  m4 r;
  for (int j = 0; j < 4; ++j) {
    const float* b = &m2->c[j].x;
    r.c[j] = add4(add4(add4(scl4(m1->c[0], b[0]), scl4(m1->c[1], b[1])), scl4(m1->c[2], b[2])), scl4(m1->c[3], b[3]));
  }
  return r;
}
/* func_matrix.inl:521-581 (inverse of mat4) */
static m4 m4_inverse(const m4* pm) {
  const m4 m = *pm;
  float Coef00 = M(m,2,2) * M(m,3,3) - M(m,3,2) * M(m,2,3);
  float Coef02 = M(m,1,2) * M(m,3,3) - M(m,3,2) * M(m,1,3);
  float Coef03 = M(m,1,2) * M(m,2,3) - M(m,2,2) * M(m,1,3);
  float Coef04 = M(m,2,1) * M(m,3,3) - M(m,3,1) * M(m,2,3);
  float Coef06 = M(m,1,1) * M(m,3,3) - M(m,3,1) * M(m,1,3);
  float Coef07 = M(m,1,1) * M(m,2,3) - M(m,2,1) * M(m,1,3);
  float Coef08 = M(m,2,1) * M(m,3,2) - M(m,3,1) * M(m,2,2);
  float Coef10 = M(m,1,1) * M(m,3,2) - M(m,3,1) * M(m,1,2);
  float Coef11 = M(m,1,1) * M(m,2,2) - M(m,2,1) * M(m,1,2);
  float Coef12 = M(m,2,0) * M(m,3,3) - M(m,3,0) * M(m,2,3);
  float Coef14 = M(m,1,0) * M(m,3,3) - M(m,3,0) * M(m,1,3);
  float Coef15 = M(m,1,0) * M(m,2,3) - M(m,2,0) * M(m,1,3);
  float Coef16 = M(m,2,0) * M(m,3,2) - M(m,3,0) * M(m,2,2);
  float Coef18 = M(m,1,0) * M(m,3,2) - M(m,3,0) * M(m,1,2);
  float Coef19 = M(m,1,0) * M(m,2,2) - M(m,2,0) * M(m,1,2);
  float Coef20 = M(m,2,0) * M(m,3,1) - M(m,3,0) * M(m,2,1);
  float Coef22 = M(m,1,0) * M(m,3,1) - M(m,3,0) * M(m,1,1);
  float Coef23 = M(m,1,0) * M(m,2,1) - M(m,2,0) * M(m,1,1);
  const v4 SignA = V4(+1, -1, +1, -1), SignB = V4(-1, +1, -1, +1);
  v4 Fac0 = V4(Coef00, Coef00, Coef02, Coef03);
  v4 Fac1 = V4(Coef04, Coef04, Coef06, Coef07);
  v4 Fac2 = V4(Coef08, Coef08, Coef10, Coef11);
  v4 Fac3 = V4(Coef12, Coef12, Coef14, Coef15);
  v4 Fac4 = V4(Coef16, Coef16, Coef18, Coef19);
  v4 Fac5 = V4(Coef20, Coef20, Coef22, Coef23);
  v4 Vec0 = V4(M(m,1,0), M(m,0,0), M(m,0,0), M(m,0,0));
  v4 Vec1 = V4(M(m,1,1), M(m,0,1), M(m,0,1), M(m,0,1));
  v4 Vec2 = V4(M(m,1,2), M(m,0,2), M(m,0,2), M(m,0,2));
  v4 Vec3 = V4(M(m,1,3), M(m,0,3), M(m,0,3), M(m,0,3));
  v4 Inv0 = mul4(SignA, add4(sub4(mul4(Vec1, Fac0), mul4(Vec2, Fac1)), mul4(Vec3, Fac2)));
  v4 Inv1 = mul4(SignB, add4(sub4(mul4(Vec0, Fac0), mul4(Vec2, Fac3)), mul4(Vec3, Fac4)));
  v4 Inv2 = mul4(SignA, add4(sub4(mul4(Vec0, Fac1), mul4(Vec1, Fac3)), mul4(Vec3, Fac5)));
  v4 Inv3 = mul4(SignB, add4(sub4(mul4(Vec0, Fac2), mul4(Vec1, Fac4)), mul4(Vec2, Fac5)));
  m4 Inverse; Inverse.c[0] = Inv0; Inverse.c[1] = Inv1; Inverse.c[2] = Inv2; Inverse.c[3] = Inv3;
  v4 Row0 = V4(M(Inverse,0,0), M(Inverse,1,0), M(Inverse,2,0), M(Inverse,3,0));
  float Determinant = dot4(m.c[0], Row0);
  for (int c = 0; c < 4; ++c) {
    Inverse.c[c].x /= Determinant; Inverse.c[c].y /= Determinant;
    Inverse.c[c].z /= Determinant; Inverse.c[c].w /= Determinant;
  }
  return Inverse;
}
/* gtc/matrix_transform.inl:44-82 (rotate, angle in DEGREES: GLM_FORCE_RADIANS is off) */
static m4 m4_rotate(const m4* m, float angle, v3 v) {
  float a = radiansf(angle);
  float c = cosf(a);                                 /* host-side set-up code: always libm */
  float s = sinf(a);
  v3 axis = normalize3(v);
  v3 temp = scl3(axis, 1.0f - c);
  float R[3][3];
  R[0][0] = c + temp.x * axis.x;
  R[0][1] = 0 + temp.x * axis.y + s * axis.z;
  R[0][2] = 0 + temp.x * axis.z - s * axis.y;
  R[1][0] = 0 + temp.y * axis.x - s * axis.z;
  R[1][1] = c + temp.y * axis.y;
  R[1][2] = 0 + temp.y * axis.z + s * axis.x;
  R[2][0] = 0 + temp.z * axis.x + s * axis.y;
  R[2][1] = 0 + temp.z * axis.y - s * axis.x;
  R[2][2] = c + temp.z * axis.z;
  m4 r;
  for (int j = 0; j < 3; ++j)
    r.c[j] = add4(add4(scl4(m->c[0], R[j][0]), scl4(m->c[1], R[j][1])), scl4(m->c[2], R[j][2]));
  r.c[3] = m->c[3];
  return r;
}
/* gtc/matrix_transform.inl:85-97 */
static m4 m4_scale(const m4* m, v3 v) {
  m4 r; r.c[0] = scl4(m->c[0], v.x); r.c[1] = scl4(m->c[1], v.y); r.c[2] = scl4(m->c[2], v.z); r.c[3] = m->c[3];
  return r;
}
typedef struct { float x, y, z, w; } quat;
/* gtc/quaternion.inl:770-790 (angleAxis, degrees) */
static quat angleAxis(float angle, v3 v) {
  quat q; float a = radiansf(angle);
  float s = sinf(a * 0.5f);                          /* host-side set-up code: always libm */
  q.w = cosf(a * 0.5f);
  q.x = v.x * s; q.y = v.y * s; q.z = v.z * s;
  return q;
}
/* gtc/quaternion.inl:626-654 (mat3_cast -> mat4_cast); gtx/quaternion.hpp:155 toMat4 */
static m4 toMat4(quat q) {
  m4 r = m4_identity();
  M(r,0,0) = 1 - 2 * q.y * q.y - 2 * q.z * q.z;
  M(r,0,1) = 2 * q.x * q.y + 2 * q.w * q.z;
  M(r,0,2) = 2 * q.x * q.z - 2 * q.w * q.y;
  M(r,1,0) = 2 * q.x * q.y - 2 * q.w * q.z;
  M(r,1,1) = 1 - 2 * q.x * q.x - 2 * q.z * q.z;
  M(r,1,2) = 2 * q.y * q.z + 2 * q.w * q.x;
  M(r,2,0) = 2 * q.x * q.z + 2 * q.w * q.y;
  M(r,2,1) = 2 * q.y * q.z - 2 * q.w * q.x;
  M(r,2,2) = 1 - 2 * q.x * q.x - 2 * q.y * q.y;
  return r;
}

/* ---- Object.inl:44-89 --------------------------------------------------------------- */
static Material material_default(void) { Material m = { { 0.5f, 0.5f, 0.5f }, 0.0f, 1.0f, 1.0f, DIFF }; return m; } /* Material.inl:22-27 */
static Object newObject(Mesh* mesh) {                                /* Object.inl:44-52 */
  Object o; memset(&o, 0, sizeof o);
  o.matrix = m4_identity(); o.matrixi = m4_identity(); o.mesh = mesh; o.material = material_default();
  return o;
}
static void obj_translate(Object* o, v3 a) {                         /* Object.inl:56-59 */
  o->matrix.c[3] = add4(o->matrix.c[3], V4(a.x, a.y, a.z, 0.0f));
  o->matrixi = m4_inverse(&o->matrix);
}
static void obj_rotate(Object* o, quat q) {                          /* Object.inl:61-65 */
  m4 r = toMat4(q);
  o->matrix = m4_mul(&r, &o->matrix);
  o->matrixi = m4_inverse(&o->matrix);
}
static void obj_scale(Object* o, v3 a) {                             /* Object.inl:81-89 */
  o->matrix = m4_scale(&o->matrix, a);
  o->matrixi = m4_inverse(&o->matrix);
}

/* ---- Mesh.h:6-18, :20-126 ------------------------------------------------------------ */
static Mesh* newGeometry(int type) {
  Mesh* m = (Mesh*)calloc(1, sizeof(Mesh));
  m->type = type; m->bmin = v3s(-0.5f); m->bmax = v3s(0.5f);
  return m;
}
static void mesh_bounds(Mesh* m) {                                   /* Mesh.h:103-113 */
  m->bmin = v3s(FLT_MAX);
  m->bmax = v3s(FLT_MIN);                                            /* sic */
  for (uint i = 0; i < m->numVerts; ++i) {
    m->bmin.x = gmin(m->bmin.x, m->verts[i].x); m->bmin.y = gmin(m->bmin.y, m->verts[i].y); m->bmin.z = gmin(m->bmin.z, m->verts[i].z);
    m->bmax.x = gmax(m->bmax.x, m->verts[i].x); m->bmax.y = gmax(m->bmax.y, m->verts[i].y); m->bmax.z = gmax(m->bmax.z, m->verts[i].z);
  }
}
static Mesh* loadObj(const char* dir, const char* filename) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s", dir, filename);
  FILE* f = fopen(path, "r");
  if (!f) {
    const char* base = strrchr(filename, '/'); base = base ? base + 1 : filename;
    snprintf(path, sizeof path, "%s/%s", dir, base);
    f = fopen(path, "r");
  }
  if (!f) { fprintf(stderr, "ERROR: loading obj:(%s)\n", path); return NULL; }
  uint cv = 64, cn = 64, cf = 64;
  Mesh* m = (Mesh*)calloc(1, sizeof(Mesh));
  m->type = MESH;
  m->verts = (v3*)malloc(cv * sizeof(v3)); m->norms = (v3*)malloc(cn * sizeof(v3)); m->faces = (Face*)malloc(cf * sizeof(Face));
  char buffer[256]; float f1, f2, f3;
  while (fgets(buffer, 255, f)) {
    if (buffer[0] == 'v' && buffer[1] == ' ') {                       /* Mesh.h:42-50 */
      if (sscanf(buffer, "v %f %f %f", &f1, &f2, &f3) == 3) {
        if (m->numVerts == cv) { cv *= 2; m->verts = (v3*)realloc(m->verts, cv * sizeof(v3)); }
        m->verts[m->numVerts++] = V3(f1, f2, f3);
      }
    }
    if (buffer[0] == 'v' && buffer[1] == 'n' && buffer[2] == ' ') {   /* Mesh.h:53-61 */
      if (sscanf(buffer, "vn %f %f %f", &f1, &f2, &f3) == 3) {
        if (m->numNorms == cn) { cn *= 2; m->norms = (v3*)realloc(m->norms, cn * sizeof(v3)); }
        m->norms[m->numNorms++] = V3(f1, f2, f3);
      }
    } else if (buffer[0] == 'f' && buffer[1] == ' ') {                /* Mesh.h:64-79 */
      Face fc;
      int nt = sscanf(buffer, "f %u//%u %u//%u %u//%u", &fc.v[0], &fc.n[0], &fc.v[1], &fc.n[1], &fc.v[2], &fc.n[2]);
      if (nt != 6) { fprintf(stderr, "ERROR: FaceMtl format\n"); fclose(f); return NULL; }
      for (int k = 0; k < 3; ++k) { fc.v[k] -= 1; fc.n[k] -= 1; }
      if (m->numFaces == cf) { cf *= 2; m->faces = (Face*)realloc(m->faces, cf * sizeof(Face)); }
      m->faces[m->numFaces++] = fc;
    }
  }
  fclose(f);
  mesh_bounds(m);
  return m;
}

/* ---- Mesh.inl:50-65 ------------------------------------------------------------------ */
static inline Triangle getTriangle(const Mesh* mesh, uint idx) {
  Triangle tri; memset(&tri, 0, sizeof tri);
  if (idx > mesh->numFaces - 1) return tri;
  tri.v[0] = mesh->verts[mesh->faces[idx].v[0]];
  tri.v[1] = mesh->verts[mesh->faces[idx].v[1]];
  tri.v[2] = mesh->verts[mesh->faces[idx].v[2]];
  tri.n[0] = mesh->norms[mesh->faces[idx].n[0]];
  tri.n[1] = mesh->norms[mesh->faces[idx].n[1]];
  tri.n[2] = mesh->norms[mesh->faces[idx].n[2]];
  return tri;
}

/* ---- Ray.inl ------------------------------------------------------------------------- */
static inline Hit Hit0(void) { Hit h; h.t = -1.0f; h.id = -1; h.pos = v3s(0.0f); h.nor = v3s(0.0f); return h; } /* Ray.inl:30 */
static inline v3 xyz(v4 v) { return V3(v.x, v.y, v.z); }

static inline Ray transformRay(Ray ray, const m4* m) {                /* Ray.inl:50-57 */
  v3 pos = xyz(m4_mulv(m, V4(ray.pos.x, ray.pos.y, ray.pos.z, 1.0f)));
  v3 pd = add3(ray.pos, ray.dir);
  v3 point = xyz(m4_mulv(m, V4(pd.x, pd.y, pd.z, 1.0f)));
  Ray r; r.pos = pos; r.dir = normalize3(sub3(point, pos));
  return r;
}
static inline Hit transformHit(Hit hit, const m4* m) {                /* Ray.inl:59-64 */
  Hit h = hit;
  h.pos = xyz(m4_mulv(m, V4(hit.pos.x, hit.pos.y, hit.pos.z, 1.0f)));
  h.nor = normalize3(xyz(m4_mulv(m, V4(hit.nor.x, hit.nor.y, hit.nor.z, 0.0f))));
  return h;
}

static inline Hit intersectTri(Ray ray, const Triangle* tri) {        /* Ray.inl:189-226 */
  v3 e1 = sub3(tri->v[1], tri->v[0]);
  v3 e2 = sub3(tri->v[2], tri->v[0]);
  v3 pvec = cross3(ray.dir, e2);
  float det = dot3(e1, pvec);
  if (gabs(det) < R_EPS) return Hit0();
  det = 1.0f / det;
  v3 tvec = sub3(ray.pos, tri->v[0]);
  v3 bary;
  bary.x = dot3(tvec, pvec) * det;
  if (bary.x < 0.0f || bary.x > 1.0f) return Hit0();
  v3 qvec = cross3(tvec, e1);
  bary.y = dot3(ray.dir, qvec) * det;
  if (bary.y < 0.0f || (bary.x + bary.y) > 1.0f) return Hit0();
  Hit hit = Hit0();
  hit.t = dot3(e2, qvec) * det;
  bary.z = 1.0f - bary.x - bary.y;                                    /* TRI_NORM_INTERP */
  hit.nor = add3(add3(scl3(tri->n[0], bary.z), scl3(tri->n[1], bary.x)), scl3(tri->n[2], bary.y));
  if (dot3(ray.dir, hit.nor) > 0) return Hit0();                      /* BACKFACE_CULL */
  return hit;
}

static inline Hit intersectMesh(Ray ray, const Mesh* mesh) {          /* Ray.inl:102-187 */
  if (mesh->type == SPHERE) {                                         /* :103-118 */
    v3 l = neg3(ray.pos);
    float s = dot3(l, ray.dir);
    float l2 = dot3(l, l);
    if (s < 0.0f && l2 > 0.25f) return Hit0();
    float m2 = l2 - s * s;
    if (m2 > 0.25f) return Hit0();
    float q = sqrtf(0.25f - m2);
    Hit hit = Hit0();
    if (l2 > 0.25f) hit.t = s - q; else hit.t = s + q;
    hit.pos = add3(ray.pos, scl3(ray.dir, hit.t));
    hit.nor = normalize3(hit.pos);
    return hit;
  } else if (mesh->type == CUBE) {                                    /* :120-140 */
    v3 tMin = div3(sub3(v3s(-0.5f), ray.pos), ray.dir);
    v3 tMax = div3(sub3(v3s(0.5f), ray.pos), ray.dir);
    v3 t1 = min3(tMin, tMax);
    v3 t2 = max3(tMin, tMax);
    float tNear = gmax(gmax(t1.x, t1.y), t1.z);
    float tFar = gmin(gmin(t2.x, t2.y), t2.z);
    if (tNear > tFar || tFar < 0.0f) return Hit0();
    Hit hit = Hit0();
    hit.t = tNear > 0.0f ? tNear : tFar;
    hit.pos = add3(ray.pos, scl3(ray.dir, hit.t));
    if      (hit.pos.x < -0.5f + R_EPS) hit.nor = V3(-1.0f, 0.0f, 0.0f);
    else if (hit.pos.x >  0.5f - R_EPS) hit.nor = V3( 1.0f, 0.0f, 0.0f);
    else if (hit.pos.y < -0.5f + R_EPS) hit.nor = V3( 0.0f, -1.0f, 0.0f);
    else if (hit.pos.y >  0.5f - R_EPS) hit.nor = V3( 0.0f, 1.0f, 0.0f);
    else if (hit.pos.z < -0.5f + R_EPS) hit.nor = V3( 0.0f, 0.0f, -1.0f);
    else                                hit.nor = V3( 0.0f, 0.0f, 1.0f);
    return hit;
  } else if (mesh->type == PLANE) {                                   /* :142-158 */
    Hit hit = Hit0();
    hit.t = -ray.pos.y / ray.dir.y;
    hit.pos = add3(ray.pos, scl3(ray.dir, hit.t));
    if (hit.pos.x < 0.5 && hit.pos.z < 0.5 && hit.pos.x > -0.5 && hit.pos.z > -0.5) {
      hit.nor = V3(0.0f, 1.0f, 0.0f);
      return hit;
    }
    return Hit0();
  } else {                                                            /* MESH, :160-186 */
    v3 tMin = div3(sub3(mesh->bmin, ray.pos), ray.dir);
    v3 tMax = div3(sub3(mesh->bmax, ray.pos), ray.dir);
    v3 t1 = min3(tMin, tMax);
    v3 t2 = max3(tMin, tMax);
    float tNear = gmax(gmax(t1.x, t1.y), t1.z);
    float tFar = gmin(gmin(t2.x, t2.y), t2.z);
    if (tNear > tFar || tFar < 0.0f) return Hit0();
    Hit hit = Hit0();
    hit.t = FLT_MAX;
    for (uint i = 0; i < mesh->numFaces; ++i) {
      Triangle tri = getTriangle(mesh, i);
      Hit thit = intersectTri(ray, &tri);
      if (thit.t > 0.0f && thit.t < hit.t) hit = thit;
    }
    hit.pos = add3(ray.pos, scl3(ray.dir, hit.t));
    return hit;
  }
}

static inline Hit intersectObject(Ray ray, const Object* obj) {       /* Ray.inl:86-100 */
  Ray r = transformRay(ray, &obj->matrixi);
  Hit hit = intersectMesh(r, obj->mesh);
  if (hit.t < 0.0f) return Hit0();
  return transformHit(hit, &obj->matrix);
}

static inline Hit intersectScene(Ray ray, const Object* scene, uint size) { /* Ray.inl:66-84 */
  float mindist = FLT_MAX;
  Hit minhit = Hit0();
  for (uint i = 0; i < size; ++i) {
    Hit h = intersectObject(ray, &scene[i]);
    if (h.t > 0.0f) {
      v3 subtract = sub3(ray.pos, h.pos);
      float dist = dot3(subtract, subtract);
      if (dist < mindist) { mindist = dist; minhit = h; minhit.id = (int)i; }
    }
  }
  return minhit;
}

/* ---- Utils.inl ----------------------------------------------------------------------- */
static inline v3 randVector(float rand1, float rand2) {               /* Utils.inl:15-23 */
  float phi = rand1 * 2.0f * R_M_PI;
  float theta = m_acos(rand2 * 2.0f - 1.0f);
  return V3(m_sin(theta) * m_cos(phi), m_sin(theta) * m_sin(phi), m_cos(theta));
}
static inline v2 randPointDisk(float rand1, float rand2, float rand3) { /* Utils.inl:25-30 */
  float t = 2.0f * R_M_PI * rand1;
  float u = rand2 + rand3;
  float r = u > 1.0f ? 2.0f - u : u;
  v2 o = { r * m_cos(t), r * m_sin(t) };
  return o;
}
static inline v3 randVectorHem(float rand1, float rand2, v3 nor) {    /* Utils.inl:32-37 */
  v3 v = randVector(rand1, rand2);
  if (dot3(v, nor) < 0) v = neg3(v);
  return v;
}

/* ---- Material.inl -------------------------------------------------------------------- */
static inline float reflectance(v3 inc, v3 nor, float n1, float n2) { /* Material.inl:58-67 */
  float n = n1 / n2;
  float cosI = -dot3(nor, inc);
  float sinT2 = n * n * (1.0f - cosI * cosI);
  if (sinT2 > 1.0f) return 1.0f;
  float cosT = sqrtf(1.0f - sinT2);
  float rOrth = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
  float rPar = (n2 * cosI - n1 * cosT) / (n2 * cosI + n1 * cosT);
  return (rOrth * rOrth + rPar * rPar) / 2.0f;
}
static inline v3 bounce(const Material* mat, v3 rd, v3 nor, v3 randvec) { /* Material.inl:69-106 */
  if (mat->type == DIFF) return randVectorHem(randvec.x, randvec.y, nor);
  else if (mat->type == MIRR) return reflect3(rd, nor);
  else {
    float n1, n2; v3 nnor;
    if (dot3(rd, nor) < 0.0f) { n1 = 1.0f; n2 = mat->n; nnor = nor; }
    else { n1 = mat->n; n2 = 1.0f; nnor = neg3(nor); }
    float refl = reflectance(rd, nnor, n1, n2);
    if (randvec.x < refl) return reflect3(rd, nnor);
    else return refract3(rd, nnor, n1 / n2);
  }
}

/* ---- raytrace.cu:21-27 --------------------------------------------------------------- */
static inline uint32_t rgbToInt(v3 c) {
  c.x = gclamp(c.x, 0.0f, 1.0f); c.y = gclamp(c.y, 0.0f, 1.0f); c.z = gclamp(c.z, 0.0f, 1.0f);
  return (uint32_t)(((int)(c.x * 255.0) << 16) | ((int)(c.y * 255.0) << 8) | (int)(c.z * 255.0));
}

/* ======================================================================================= */
/* API                                                                                     */
/* ======================================================================================= */
typedef struct { Object* objs; uint n; } Scene;

const char* orc_name(void) { return "c-restatement"; }
int orc_set_math_mode(int mode) { if (mode != 0 && mode != 1) return -1; g_math_mode = mode; return 0; }

/* main.cpp:281-337: five walls and the ceiling light */
static uint wallsAndLight(Object* o) {
  const v3 BOX_HDIM = v3s(5.0f);
  const v3 X = V3(1.0f, 0.0f, 0.0f), Z = V3(0.0f, 0.0f, 1.0f);
  Mesh* planemesh = newGeometry(PLANE);
  uint k = 0;
  /* bottom -y, main.cpp:288-292 */
  o[k] = newObject(planemesh);
  obj_scale(&o[k], scl3(BOX_HDIM, 2.0f));
  obj_translate(&o[k], V3(0.0f, -BOX_HDIM.y, 0.0f));
  o[k].material.color = v3s(1.0f); ++k;
  /* top +y, :294-300 */
  o[k] = newObject(planemesh);
  obj_rotate(&o[k], angleAxis(180.0f, X));
  obj_scale(&o[k], scl3(BOX_HDIM, 2.0f));
  obj_translate(&o[k], V3(0.0f, BOX_HDIM.y, 0.0f));
  o[k].material.color = V3(1.0f, 1.0f, 1.0f); ++k;
  /* back -z, :302-307 */
  o[k] = newObject(planemesh);
  obj_rotate(&o[k], angleAxis(90.0f, X));
  obj_scale(&o[k], scl3(BOX_HDIM, 2.0f));
  obj_translate(&o[k], V3(0.0f, 0.0f, -BOX_HDIM.z));
  o[k].material.color = v3s(1.0f); ++k;
  /* right +x, :316-322 */
  o[k] = newObject(planemesh);
  obj_rotate(&o[k], angleAxis(90.0f, Z));
  obj_scale(&o[k], scl3(BOX_HDIM, 2.0f));
  obj_translate(&o[k], V3(BOX_HDIM.x, 0.0f, 0.0f));
  o[k].material.color = V3(0.5f, 0.5f, 1.0f); ++k;
  /* left -x, :324-329 */
  o[k] = newObject(planemesh);
  obj_rotate(&o[k], angleAxis(-90.0f, Z));
  obj_scale(&o[k], scl3(BOX_HDIM, 2.0f));
  obj_translate(&o[k], V3(-BOX_HDIM.x, 0.0f, 0.0f));
  o[k].material.color = V3(1.0f, 0.5f, 0.5f); ++k;
  /* ceiling light, :332-337 */
  o[k] = newObject(newGeometry(CUBE));
  obj_scale(&o[k], V3(BOX_HDIM.x, 0.5f, BOX_HDIM.z));
  obj_translate(&o[k], V3(0.0f, BOX_HDIM.y, 0.0f));
  o[k].material.color = V3(1.0f, 1.0f, 1.0f);
  o[k].material.emit = 2.0f; ++k;
  return k;
}

void* orc_scene_create(int variant, int param, const char* data_dir) {
  Scene* s = (Scene*)calloc(1, sizeof(Scene));
  s->objs = (Object*)calloc(16, sizeof(Object));
  Object* o = s->objs;
  uint k = wallsAndLight(o);
  if (variant == 0 || variant == 1) {
    /* CORNELL BOX SPHERES, main.cpp:370-385 */
    o[k] = newObject(newGeometry(SPHERE));
    obj_scale(&o[k], v3s(4.0f));
    obj_translate(&o[k], V3(-2.5f, -3.0f, -3.0f));
    o[k].material.color = V3(1.0f, 1.0f, 1.0f);
    o[k].material.type = MIRR; ++k;
    o[k] = newObject(newGeometry(SPHERE));
    obj_scale(&o[k], v3s(4.0f));
    obj_translate(&o[k], V3(2.5f, -3.0f, 0.0f));
    o[k].material.color = V3(1.0f, 1.0f, 1.0f);
    o[k].material.type = TRANS;
    o[k].material.n = 1.6f; ++k;
    /* icosahedron, main.cpp:387-395 */
    Mesh* ico = loadObj(data_dir ? data_dir : ".", "data/icosahedron.obj");
    if (!ico) { free(s->objs); free(s); return NULL; }
    o[k] = newObject(ico);
    obj_scale(&o[k], v3s(1.5f));
    obj_rotate(&o[k], angleAxis(-10.0f, V3(1.0f, 0.0f, 0.0f)));
    obj_rotate(&o[k], angleAxis(25.0f, V3(0.0f, 1.0f, 0.0f)));
    obj_translate(&o[k], V3(-1.0f, -3.7f, 3.0f));
    o[k].material.color = V3(1.0f, 1.0f, 0.0f); ++k;
    if (variant == 1) { o[8].material.type = TRANS; o[8].material.n = 1.6f; }
  } else if (variant == 2) {
    Mesh* base = loadObj(data_dir ? data_dir : ".", "data/icosahedron.obj");
    if (!base) { free(s->objs); free(s); return NULL; }
    uint32_t* f0 = (uint32_t*)malloc(sizeof(uint32_t) * 3 * base->numFaces);
    for (uint i = 0; i < base->numFaces; ++i) for (int c = 0; c < 3; ++c) f0[3*i+c] = base->faces[i].v[c];
    ico_mesh im = ico_subdivide((const float*)base->verts, base->numVerts, f0, base->numFaces, param);
    free(f0);
    Mesh* m = (Mesh*)calloc(1, sizeof(Mesh));
    m->type = MESH;
    m->numVerts = im.nverts; m->numNorms = im.nverts; m->numFaces = im.nfaces;
    m->verts = (v3*)malloc(sizeof(v3) * im.nverts); memcpy(m->verts, im.verts, sizeof(v3) * im.nverts);
    m->norms = (v3*)malloc(sizeof(v3) * im.nverts); memcpy(m->norms, im.verts, sizeof(v3) * im.nverts);
    const float radius = ico_radius(param);                       /* icosphere.h: keeps |det| above EPS */
    for (uint i = 0; i < im.nverts; ++i) { m->verts[i].x *= radius; m->verts[i].y *= radius; m->verts[i].z *= radius; }
    m->faces = (Face*)malloc(sizeof(Face) * im.nfaces);
    for (uint i = 0; i < im.nfaces; ++i) for (int c = 0; c < 3; ++c) { m->faces[i].v[c] = im.faces[3*i+c]; m->faces[i].n[c] = im.faces[3*i+c]; }
    free(im.verts); free(im.faces);
    mesh_bounds(m);
    o[k] = newObject(m);
    obj_scale(&o[k], v3s(3.0f / ico_radius(param)));
    obj_translate(&o[k], V3(0.0f, -2.0f, 0.0f));
    o[k].material.color = V3(1.0f, 1.0f, 0.0f); ++k;
  } else { free(s->objs); free(s); return NULL; }
  s->n = k;
  return s;
}

/* A scene built elsewhere: `objects` is an Object[] in the reference's host layout (Object.inl:24-29 with
 * m_mesh -> Mesh.inl:34-41), e.g. what the product's scene-description API hands to trc_upload_scene. The
 * records are copied, the meshes they point at are shared and must outlive the scene. */
void* orc_scene_wrap(const void* objects, unsigned n) {
  if (!objects || n == 0) return NULL;
  Scene* s = (Scene*)calloc(1, sizeof(Scene));
  s->objs = (Object*)malloc(sizeof(Object) * n);
  memcpy(s->objs, objects, sizeof(Object) * n);
  s->n = n;
  return s;
}

void orc_scene_free(void* sc) { Scene* s = (Scene*)sc; if (s) { free(s->objs); free(s); } }
unsigned orc_scene_size(const void* sc) { return ((const Scene*)sc)->n; }
const void* orc_scene_objects(const void* sc) { return ((const Scene*)sc)->objs; }

void orc_scene_object_info(const void* sc, unsigned i, float matrix[16], float matrixi[16],
                           int* mesh_type, unsigned counts[3], float bmin[3], float bmax[3],
                           float color[3], float emit_brdf_n[3], int* mat_type) {
  const Object* o = &((const Scene*)sc)->objs[i];
  memcpy(matrix, &o->matrix, 64); memcpy(matrixi, &o->matrixi, 64);
  *mesh_type = o->mesh->type;
  counts[0] = o->mesh->numVerts; counts[1] = o->mesh->numNorms; counts[2] = o->mesh->numFaces;
  memcpy(bmin, &o->mesh->bmin, 12); memcpy(bmax, &o->mesh->bmax, 12);
  memcpy(color, &o->material.color, 12);
  emit_brdf_n[0] = o->material.emit; emit_brdf_n[1] = o->material.brdf; emit_brdf_n[2] = o->material.n;
  *mat_type = o->material.type;
}
void orc_scene_object_mesh(const void* sc, unsigned i, float* verts, float* norms, unsigned* faces6) {
  const Mesh* m = ((const Scene*)sc)->objs[i].mesh;
  if (verts) memcpy(verts, m->verts, 12ull * m->numVerts);
  if (norms) memcpy(norms, m->norms, 12ull * m->numNorms);
  if (faces6) memcpy(faces6, m->faces, 24ull * m->numFaces);
}
void orc_scene_translate(void* sc, unsigned i, const float a[3]) { obj_translate(&((Scene*)sc)->objs[i], V3(a[0], a[1], a[2])); }
void orc_scene_rotate(void* sc, unsigned i, float angle_deg, const float ax[3]) { obj_rotate(&((Scene*)sc)->objs[i], angleAxis(angle_deg, V3(ax[0], ax[1], ax[2]))); }
void orc_scene_scale(void* sc, unsigned i, const float a[3]) { obj_scale(&((Scene*)sc)->objs[i], V3(a[0], a[1], a[2])); }

/* Camera.cpp:3-20 (ctor), :79-112 (ThirdPersonCamera), main.cpp:275-279, Renderer.cpp:156-171 */
void orc_camera_default(unsigned aspect_w, unsigned aspect_h, float rot_x, float rot_y, orc_camera* out) {
  float m_currXZRads = 0.0f, m_currYRads = 0.0f, m_radius = 0.0f;
  v3 m_lookAt = v3s(0.0f);                                  /* glm::vec3() zero-initialises */
  float m_fov = (float)R_FOV;                               /* setFOV(FOV) */
  float m_aspect = (float)aspect_w / aspect_h;              /* Camera.cpp:33 */
  m_radius -= -13.0f;                                       /* zoom(-13), Camera.cpp:93 */
  float focalDist = 13.0f, lensRadius = 1.0f;               /* main.cpp:278, Camera.cpp:17 */
  m_currXZRads += rot_x; m_currYRads += rot_y;              /* Camera.cpp:81-82 */
  /* update(), Camera.cpp:96-112 */
  float cosa = cosf(m_currXZRads);
  float sina = sinf(m_currXZRads);
  v3 currPos = V3(sina, 0.0f, cosa);
  v3 UpRotAxis = V3(currPos.z, currPos.y, -currPos.x);
  m4 I = m4_identity();
  m4 xRotation = m4_rotate(&I, degreesf(m_currYRads), UpRotAxis);
  currPos = xyz(m4_mulv(&xRotation, V4(currPos.x, currPos.y, currPos.z, 0.0f)));
  v3 tempVec = scl3(currPos, m_radius);
  v3 m_position = add3(tempVec, m_lookAt);
  v3 m_upDir = normalize3(cross3(currPos, UpRotAxis));
  /* Renderer.cpp:156-171 */
  v3 Cc = normalize3(sub3(m_lookAt, m_position));
  v3 A = normalize3(cross3(Cc, m_upDir));
  v3 B = scl3(normalize3(cross3(A, Cc)), 1.0f / m_aspect);
  float tanFOV = tanf(radiansf(m_fov));                     /* Renderer.cpp:168: with GLM's headers in the TU
                                                               `tan(float)` resolves to the float overload */
  A = scl3(A, tanFOV);
  B = scl3(B, tanFOV);
  memcpy(out->campos, &m_position, 12); memcpy(out->A, &A, 12); memcpy(out->B, &B, 12); memcpy(out->C, &Cc, 12);
  out->lensRadius = lensRadius; out->focalDist = focalDist;
}

static inline v3 ldv(const float* p) { return V3(p[0], p[1], p[2]); }

/* raytraceKernel, raytrace.cu:29-62 */
void orc_raytrace1(const void* sc, unsigned w, unsigned h, const orc_camera* cam,
                   uint32_t* pbo, int* ids, float* ts, float* pos3, float* nor3, int nthreads) {
  const Scene* s = (const Scene*)sc;
  const v3 campos = ldv(cam->campos), A = ldv(cam->A), B = ldv(cam->B), Cc = ldv(cam->C);
  const uint size = w * h;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  #pragma omp parallel for schedule(dynamic, 256)
  for (uint idx = 0; idx < size; ++idx) {
    uint x = idx % w;
    uint y = idx / w;
    v2 uv = { (float)x / w, (float)y / h };
    Ray ray;
    ray.pos = add3(add3(add3(campos, Cc), scl3(A, 2.0f * uv.x - 1.0f)), scl3(B, 2.0f * uv.y - 1.0f));
    ray.dir = normalize3(sub3(ray.pos, campos));
    v3 lightDir = V3((float)0.267261, (float)0.801784, (float)0.534522);
    Hit hit = intersectScene(ray, s->objs, s->n);
    v3 col;
    if (hit.id < 0) col = ray.dir;
    else {
      const Material* mt = &s->objs[hit.id].material;
      if (mt->emit > 0.0f) col = mt->color;
      else {
        col = scl3(mt->color, mt->brdf);
        col = scl3(col, gmax(dot3(lightDir, hit.nor), 0.0f));
      }
    }
    if (pbo) pbo[idx] = rgbToInt(col);
    if (ids) ids[idx] = hit.id;
    if (ts) ts[idx] = hit.t;
    if (pos3) memcpy(pos3 + 3 * idx, &hit.pos, 12);
    if (nor3) memcpy(nor3 + 3 * idx, &hit.nor, 12);
  }
}

/* Where Ray.inl:189-226 leaves for the (ray, face) pairs of a mode-1 frame: the primary rays of every `step`-th pixel
 * in x and y (raytrace.cu:40-43) against MESH object `obj`; a ray that passes the AABB (Ray.inl:161-167) meets every
 * face (:172). counts[0..3] = pairs that return at :196 (|det| < EPS), :203 (bary.x), :208 (bary.y), and pairs that
 * run the whole test (:209-225); counts[4] = rays that passed the AABB, counts[5] = rays tried. SURVEY.md 8(d) prices
 * the four stages at 20 / 30 / 46 / 74 flop. */
void orc_mesh_exit_histogram(const void* sc, unsigned w, unsigned h, const orc_camera* cam, unsigned obj, unsigned step,
                             uint64_t counts[6], int nthreads) {
  const Scene* s = (const Scene*)sc;
  const v3 campos = ldv(cam->campos), A = ldv(cam->A), B = ldv(cam->B), Cc = ldv(cam->C);
  const Object* o = &s->objs[obj];
  const Mesh* mesh = o->mesh;
  uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, pass = 0, tried = 0;
  if (step == 0) step = 1;
  const uint nx = (w + step - 1) / step, ny = (h + step - 1) / step;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  #pragma omp parallel for schedule(dynamic, 1) reduction(+: c0, c1, c2, c3, pass, tried)
  for (uint k = 0; k < nx * ny; ++k) {
    const uint x = (k % nx) * step, y = (k / nx) * step;
    v2 uv = { (float)x / w, (float)y / h };
    Ray wray;
    wray.pos = add3(add3(add3(campos, Cc), scl3(A, 2.0f * uv.x - 1.0f)), scl3(B, 2.0f * uv.y - 1.0f));
    wray.dir = normalize3(sub3(wray.pos, campos));
    const Ray ray = transformRay(wray, &o->matrixi);
    ++tried;
    v3 tMin = div3(sub3(mesh->bmin, ray.pos), ray.dir);
    v3 tMax = div3(sub3(mesh->bmax, ray.pos), ray.dir);
    v3 t1 = min3(tMin, tMax);
    v3 t2 = max3(tMin, tMax);
    float tNear = gmax(gmax(t1.x, t1.y), t1.z);
    float tFar = gmin(gmin(t2.x, t2.y), t2.z);
    if (tNear > tFar || tFar < 0.0f) continue;
    ++pass;
    for (uint i = 0; i < mesh->numFaces; ++i) {
      Triangle tri = getTriangle(mesh, i);
      v3 e1 = sub3(tri.v[1], tri.v[0]);
      v3 e2 = sub3(tri.v[2], tri.v[0]);
      v3 pvec = cross3(ray.dir, e2);
      float det = dot3(e1, pvec);
      if (gabs(det) < R_EPS) { ++c0; continue; }
      det = 1.0f / det;
      v3 tvec = sub3(ray.pos, tri.v[0]);
      float bx = dot3(tvec, pvec) * det;
      if (bx < 0.0f || bx > 1.0f) { ++c1; continue; }
      v3 qvec = cross3(tvec, e1);
      float by = dot3(ray.dir, qvec) * det;
      if (by < 0.0f || (bx + by) > 1.0f) { ++c2; continue; }
      ++c3;
    }
  }
  counts[0] = c0; counts[1] = c1; counts[2] = c2; counts[3] = c3; counts[4] = pass; counts[5] = tried;
}

static inline v3 rand3v(uint64_t seed, uint32_t index, uint32_t iter, uint32_t dim) {
  float r[3]; orc_rand3(seed, index, iter, dim, r); return V3(r[0], r[1], r[2]);
}

/* pathtrace(), raytrace.cu:190-226, with the reference's buffer structure */
void orc_pathtrace(const void* sc, unsigned w, unsigned h, const orc_camera* cam,
                   uint64_t seed, unsigned iter_first, unsigned iter_stride, unsigned n_iters,
                   int rng_mode, float* filmF, uint32_t* pbo, orc_stats* stats, int nthreads) {
  const Scene* s = (const Scene*)sc;
  const Object* scene = s->objs; const uint sceneSize = s->n;
  const v3 campos = ldv(cam->campos), A = ldv(cam->A), B = ldv(cam->B), Cc = ldv(cam->C);
  const float lensRadius = cam->lensRadius, focalDist = cam->focalDist;
  const uint size = w * h;
  Ray* rays = (Ray*)malloc(sizeof(Ray) * size);
  v3* col = (v3*)malloc(sizeof(v3) * size);
  int* indices = (int*)malloc(sizeof(int) * size);
  v3* film = (v3*)filmF;
  uint64_t c_live = 0, c_miss = 0, c_light = 0, c_cap = 0, c_diff = 0, c_trans = 0, c_mirr = 0, c_paths = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif

  for (unsigned k = 0; k < n_iters; ++k) {
    const uint filmIters = iter_first + k * iter_stride;

    /* initBuffersKernel, raytrace.cu:64-105; rand[idx] := rand3(seed, idx, filmIters, 0) */
    #pragma omp parallel for schedule(static)
    for (uint idx = 0; idx < size; ++idx) {
      uint x = idx % w;
      uint y = idx / w;
      v3 r = rand3v(seed, idx, filmIters, 0);
      v2 uv = { (float)x / w + (2.0f * r.x - 1.0f) / w,
                (float)y / h + (2.0f * r.y - 1.0f) / h };              /* PIX_JITTER :80-83 */
      rays[idx].pos = add3(add3(add3(campos, Cc), scl3(A, 2.0f * uv.x - 1.0f)), scl3(B, 2.0f * uv.y - 1.0f));
      rays[idx].dir = normalize3(sub3(rays[idx].pos, campos));
      v3 fpt = add3(scl3(rays[idx].dir, focalDist), rays[idx].pos);    /* FOCAL_BLUR :91-95 */
      v2 rd = randPointDisk(r.x, r.y, r.z);
      v2 randdisk = { lensRadius * rd.x, lensRadius * rd.y };
      rays[idx].pos = add3(rays[idx].pos, add3(scl3(normalize3(A), randdisk.y), scl3(normalize3(B), randdisk.x)));
      rays[idx].dir = normalize3(sub3(fpt, rays[idx].pos));
      col[idx] = v3s(1.0f);
      indices[idx] = (int)idx;
      if (filmIters == 1) film[idx] = v3s(0.0f);
    }
    c_paths += size;

    /* calcColorKernel x PATH_DEPTH, raytrace.cu:107-160, :217-221 */
    for (int depth = 0; depth < R_PATH_DEPTH; ++depth) {
      #pragma omp parallel for schedule(dynamic, 256) reduction(+:c_live,c_miss,c_light,c_cap,c_diff,c_trans,c_mirr)
      for (uint idx = 0; idx < size; ++idx) {
        if (indices[idx] == -1) continue;
        ++c_live;
        Hit hit = intersectScene(rays[idx], scene, sceneSize);
        if (hit.id < 0) { col[idx] = v3s(0.0f); indices[idx] = -1; ++c_miss; continue; }
        const Material* mt = &scene[hit.id].material;
        if (mt->emit > 0.0f) {
          col[idx] = mul3(col[idx], scl3(mt->color, mt->emit));
          indices[idx] = -1; ++c_light; continue;
        }
        if (depth == R_PATH_DEPTH - 1) { col[idx] = v3s(0.0f); ++c_cap; continue; }
        col[idx] = mul3(col[idx], mt->color);
        v3 rnd;
        if (rng_mode == ORC_RNG_REF) {
          uint randidx = (idx + (int)dot3(rays[idx].dir, rays[idx].pos)) % (size);   /* :155 */
          rnd = rand3v(seed, randidx, filmIters, 0);
        } else {
          rnd = rand3v(seed, idx, filmIters, 1u + (uint)depth);
        }
        if (mt->type == DIFF) ++c_diff; else if (mt->type == TRANS) ++c_trans; else ++c_mirr;
        rays[idx].dir = bounce(mt, rays[idx].dir, hit.nor, rnd);
        rays[idx].pos = add3(hit.pos, scl3(rays[idx].dir, R_EPS));
      }
    }

    /* accumColorKernel, raytrace.cu:162-177 */
    const float fIters = (float)filmIters;
    const int last = (k + 1 == n_iters);
    #pragma omp parallel for schedule(static)
    for (uint idx = 0; idx < size; ++idx) {
      film[idx] = add3(film[idx], col[idx]);
      if (pbo && last) {
        v3 g = V3(m_pow(film[idx].x / fIters, 1.0f / 2.2f), m_pow(film[idx].y / fIters, 1.0f / 2.2f), m_pow(film[idx].z / fIters, 1.0f / 2.2f));
        pbo[idx] = rgbToInt(g);
      }
    }
  }
  free(rays); free(col); free(indices);
  if (stats) {
    stats->live_segments = c_live; stats->miss = c_miss; stats->light = c_light; stats->depthcap = c_cap;
    stats->diff = c_diff; stats->trans = c_trans; stats->mirr = c_mirr; stats->paths = c_paths;
  }
}

void orc_tonemap(const float* filmF, unsigned n, float iters, uint32_t* pbo) {
  const v3* film = (const v3*)filmF;
  for (unsigned idx = 0; idx < n; ++idx) {
    v3 g = V3(m_pow(film[idx].x / iters, 1.0f / 2.2f), m_pow(film[idx].y / iters, 1.0f / 2.2f), m_pow(film[idx].z / iters, 1.0f / 2.2f));
    pbo[idx] = rgbToInt(g);
  }
}

void orc_intersect_scene(const void* sc, const float r6[6], float* t, int* id, float pos[3], float nor[3]) {
  const Scene* s = (const Scene*)sc;
  Ray ray = { ldv(r6), ldv(r6 + 3) };
  Hit h = intersectScene(ray, s->objs, s->n);
  *t = h.t; *id = h.id; memcpy(pos, &h.pos, 12); memcpy(nor, &h.nor, 12);
}
void orc_intersect_object(const void* sc, unsigned obj, const float r6[6], float* t, float pos[3], float nor[3]) {
  const Scene* s = (const Scene*)sc;
  Ray ray = { ldv(r6), ldv(r6 + 3) };
  Hit h = intersectObject(ray, &s->objs[obj]);
  *t = h.t; memcpy(pos, &h.pos, 12); memcpy(nor, &h.nor, 12);
}
/* main.cpp:180-208 */
void orc_pick(const void* sc, const orc_camera* cam, int window_w, int window_h, int x, int y,
              float ray6[6], float* t, int* id, float pos[3], float nor[3]) {
  const Scene* s = (const Scene*)sc;
  const v3 campos = ldv(cam->campos), A = ldv(cam->A), B = ldv(cam->B), Cc = ldv(cam->C);
  v2 uv = { (float)x / window_w, (float)(window_h - y) / window_h };                    /* :196 */
  Ray ray;
  ray.pos = add3(add3(add3(campos, Cc), scl3(A, 2.0f * uv.x - 1.0f)), scl3(B, 2.0f * uv.y - 1.0f));   /* :198 */
  ray.dir = normalize3(sub3(ray.pos, campos));                                          /* :199 */
  Hit h = intersectScene(ray, s->objs, s->n);                                           /* :202 */
  memcpy(ray6, &ray.pos, 12); memcpy(ray6 + 3, &ray.dir, 12);
  *t = h.t; *id = h.id; memcpy(pos, &h.pos, 12); memcpy(nor, &h.nor, 12);
}
void orc_bounce(int mat_type, float mat_n, const float rd[3], const float nor[3], const float rnd[3], float out[3]) {
  Material m = material_default(); m.type = mat_type; m.n = mat_n;
  v3 o = bounce(&m, ldv(rd), ldv(nor), ldv(rnd));
  memcpy(out, &o, 12);
}
float orc_reflectance(const float inc[3], const float nor[3], float n1, float n2) { return reflectance(ldv(inc), ldv(nor), n1, n2); }
void orc_rand_vector(float r1, float r2, float out[3]) { v3 o = randVector(r1, r2); memcpy(out, &o, 12); }
void orc_rand_vector_hem(float r1, float r2, const float nor[3], float out[3]) { v3 o = randVectorHem(r1, r2, ldv(nor)); memcpy(out, &o, 12); }
void orc_rand_point_disk(float r1, float r2, float r3, float out[2]) { v2 o = randPointDisk(r1, r2, r3); out[0] = o.x; out[1] = o.y; }
void orc_rand3_api(uint64_t seed, uint32_t index, uint32_t iter, uint32_t dim, float out[3]) { orc_rand3(seed, index, iter, dim, out); }
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32_10(ctr, key, out); }
uint32_t orc_rgb_to_int(float r, float g, float b) { return rgbToInt(V3(r, g, b)); }

void orc_pm_sincos(float x, float* s, float* c) { pm_sincosf(x, s, c); }
float orc_pm_acos(float x) { return pm_acosf(x); }
float orc_pm_pow(float x, float y) { return pm_powf(x, y); }
