/*
 * craftax_classic.c -- CPU ORACLE for Craftax-Classic-Symbolic-v1.  TEST INFRASTRUCTURE ONLY (see pqn_oracle.h).
 *
 * The env of BASELINE.json configs[4] / reference call sites purejaxql/pqn_craftax.py:96-99,202-204,433-439 lives in
 * the un-vendored third-party package craftax>=1.4.2 (reference pyproject.toml:50), which is NOT under
 * /root/reference and cannot be installed here.  **PARITY UNPINNED**: this file restates the published game rules --
 * Craftax-Classic is a JAX re-implementation of Crafter (Hafner 2021): 64x64 procedurally generated map, 17 block
 * types, 17 actions, 22 achievements, zombies / cows / skeletons / arrows, growing plants, health-food-drink-energy
 * with day / night -- from the Crafter paper and recollection of the two code bases.  Where the recollection is not
 * certain the rule chosen here is stated in the comment next to it.  World generation uses this build's own
 * fixed-point value noise (jax's PRNG streams and Craftax's Perlin noise cannot be reproduced without jax), so maps are
 * statistically Crafter-like, not sample-identical.  What IS pinned: the HIP kernels reproduce this file bit for bit
 * (tests/test_craftax_env_gpu.py), hand-derived known-answer steps (tests/test_craftax_env_cpu.py), and the
 * observation layout of the reference's network input: 7 x 9 local view x (17 block one-hots + 4 mob channels)
 * + 12 inventory + 4 intrinsics + 4 direction one-hot + light level + is_sleeping = 1345 floats.
 *
 * Canonical state per env: si[CC_SI] int32, sf[4] float:
 *   si[0..4095]            map[r*64+c] block id
 *   S = 4096:  S+0 player_r, S+1 player_c, S+2 direction (1 left, 2 right, 3 up, 4 down), S+3 health, S+4 food,
 *   S+5 drink, S+6 energy, S+7 is_sleeping, S+8..19 inventory (wood, stone, coal, iron, diamond, sapling,
 *   wood/stone/iron pickaxe, wood/stone/iron sword), S+20+5i zombie i (r, c, health, cooldown, mask), S+35+4i cow i
 *   (r, c, health, mask), S+47+5i skeleton i (r, c, health, reload, mask), S+57+4i arrow i (r, c, dir, mask),
 *   S+69+4i plant i (r, c, age, mask), S+109 achievements (bit k = achievement k), S+110 timestep
 *   sf: recover, hunger, thirst, fatigue
 * Randomness: counter-based, env e draws pqn_oracle_env_bits(key, e, stream) with the streams named below.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pqn_oracle.h"

enum { B_INVALID, B_OOB, B_GRASS, B_WATER, B_STONE, B_TREE, B_WOOD, B_PATH, B_COAL, B_IRON, B_DIAMOND, B_TABLE, B_FURNACE,
       B_SAND, B_LAVA, B_PLANT, B_RIPE, B_COUNT };
enum { A_NOOP, A_LEFT, A_RIGHT, A_UP, A_DOWN, A_DO, A_SLEEP, A_PLACE_STONE, A_PLACE_TABLE, A_PLACE_FURNACE, A_PLACE_PLANT,
       A_MAKE_WOOD_PICKAXE, A_MAKE_STONE_PICKAXE, A_MAKE_IRON_PICKAXE, A_MAKE_WOOD_SWORD, A_MAKE_STONE_SWORD,
       A_MAKE_IRON_SWORD, A_COUNT };
enum { ACH_COLLECT_COAL, ACH_COLLECT_DIAMOND, ACH_COLLECT_DRINK, ACH_COLLECT_IRON, ACH_COLLECT_SAPLING, ACH_COLLECT_STONE,
       ACH_COLLECT_WOOD, ACH_DEFEAT_SKELETON, ACH_DEFEAT_ZOMBIE, ACH_EAT_COW, ACH_EAT_PLANT, ACH_MAKE_IRON_PICKAXE,
       ACH_MAKE_IRON_SWORD, ACH_MAKE_STONE_PICKAXE, ACH_MAKE_STONE_SWORD, ACH_MAKE_WOOD_PICKAXE, ACH_MAKE_WOOD_SWORD,
       ACH_PLACE_FURNACE, ACH_PLACE_PLANT, ACH_PLACE_STONE, ACH_PLACE_TABLE, ACH_WAKE_UP, ACH_COUNT };
enum { I_WOOD, I_STONE, I_COAL, I_IRON, I_DIAMOND, I_SAPLING, I_WOOD_PICKAXE, I_STONE_PICKAXE, I_IRON_PICKAXE, I_WOOD_SWORD,
       I_STONE_SWORD, I_IRON_SWORD };
#define CC_MAP 64
#define CC_S 4096
#define CC_SI (4096 + 111)
#define CC_SF 4
#define CC_OBS 1345
#define CC_MAX_STEPS 10000
#define CC_NZ 3
#define CC_NC 3
#define CC_NS 2
#define CC_NA 3
#define CC_NP 10
/* step streams */
enum { ST_SAPLING = 10, ST_ZOMBIE = 11, ST_COW = 14, ST_SKEL_A = 17, ST_SKEL_B = 19, ST_SPAWN_COW = 21, ST_SPAWN_ZOMBIE = 22,
       ST_SPAWN_SKEL = 23, ST_WORLD = 40 };

static const int DR[5] = {0, 0, 0, -1, 1}, DC[5] = {0, -1, 1, 0, 0};

/* ---- explicit-f32 cosine (the sequence of pqn_oracle.c oracle_sincos_f32, restated) -------------------------- */
static float cc_cos(float x) {
  const float k = rintf(x * 0.636619772367581343f);
  float r = x - k * 1.5703125f;
  r = r - k * 4.837512969970703125e-4f;
  r = r - k * 7.54978995489188216e-8f;
  const float z = r * r;
  const float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  const float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  const int q = ((int)k) & 3;
  return (q == 0) ? cp : (q == 1) ? -sp : (q == 2) ? -cp : sp;
}
/* daylight of Crafter: 1 - |cos(pi * t / 300)|^3 shifted by 0.3 of a day; t = timestep */
static float cc_light(int32_t t) {
  const float progress = (float)(t % 300) / 300.0f + 0.3f;
  const float c = fabsf(cc_cos(3.14159265358979323846f * progress));
  return 1.0f - c * c * c;
}

/* ---- world generation: fixed-point value noise ---------------------------------------------------------------- */
static uint32_t cc_hash(uint32_t x) { /* lowbias32 */
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
static int32_t cc_lattice(uint32_t seed, int32_t ix, int32_t iy, uint32_t layer) { /* value in [-32768, 32767] */
  uint32_t h = cc_hash(seed ^ ((uint32_t)ix * 0x9E3779B1U) ^ ((uint32_t)iy * 0x85EBCA77U) ^ (layer * 0xC2B2AE3DU));
  return (int32_t)(h >> 16) - 32768;
}
/* noise at (x, y) given in 1/256 cells, lattice spacing `size` cells; result Q15 in [-32768, 32767] */
static int32_t cc_noise(uint32_t seed, int32_t x256, int32_t y256, uint32_t layer, int32_t size) {
  const int32_t span = size * 256;
  const int32_t ix = x256 / span, iy = y256 / span;            /* coordinates are >= 0 */
  const int32_t fx = ((x256 - ix * span) * 256) / span, fy = ((y256 - iy * span) * 256) / span; /* 0..255 */
  const int32_t sx = (fx * fx * (768 - 2 * fx)) >> 16, sy = (fy * fy * (768 - 2 * fy)) >> 16;   /* smoothstep, 0..256 */
  const int32_t v00 = cc_lattice(seed, ix, iy, layer), v10 = cc_lattice(seed, ix + 1, iy, layer);
  const int32_t v01 = cc_lattice(seed, ix, iy + 1, layer), v11 = cc_lattice(seed, ix + 1, iy + 1, layer);
  const int32_t a = v00 + (((v10 - v00) * sx) >> 8), b = v01 + (((v11 - v01) * sx) >> 8);
  return a + (((b - a) * sy) >> 8);
}
/* two octaves, weights in 1/256: (w1 n(size1) + w2 n(size2)) / (w1 + w2), Q15 */
static int32_t cc_fnoise(uint32_t seed, int32_t x256, int32_t y256, uint32_t layer, int32_t s1, int32_t w1, int32_t s2, int32_t w2) {
  const int32_t n1 = cc_noise(seed, x256, y256, layer, s1);
  if (w2 == 0) return n1;
  const int32_t n2 = cc_noise(seed, x256, y256, layer + 16u, s2);
  return (n1 * w1 + n2 * w2) / (w1 + w2);
}
static uint32_t cc_isqrt(uint32_t v) { /* floor(sqrt(v)) */
  uint32_t r = 0, bit = 1u << 30;
  while (bit > v) bit >>= 2;
  while (bit) { if (v >= r + bit) { v -= r + bit; r = (r >> 1) + bit; } else r >>= 1; bit >>= 2; }
  return r;
}
#define Q15(x) ((int32_t)((x) * 32768.0))
/* block of cell (r, c) of the world generated from `seed`: Crafter's worldgen (crafter/worldgen.py _set_material)
 * with value noise for simplex noise and a hard sigmoid for the start-area blend; u = per-cell uniform in [0, 65535] */
int32_t cc_world_cell(uint32_t seed, int32_t r, int32_t c) {
  const int32_t x = c * 256, y = r * 256;
  const uint32_t u = cc_hash(seed ^ 0xA511E9B3U ^ (uint32_t)(r * 64 + c) * 0x9E3779B1U) >> 16;
  const int32_t dx = c - 32, dy = r - 32;
  const int32_t dist_q8 = (int32_t)cc_isqrt((uint32_t)(dx * dx + dy * dy) << 16);            /* Q8 distance to the spawn */
  /* start = sigmoid(4 - dist + 2 n(8; size 3)), hard sigmoid clamp(0.5 + x / 4, 0, 1), Q15 */
  int32_t sraw = Q15(4.0) - dist_q8 * 128 + 2 * cc_noise(seed, x, y, 8u, 3);
  int32_t start = Q15(0.5) + sraw / 4;
  start = start < 0 ? 0 : (start > 32767 ? 32767 : start);
  int32_t water = cc_fnoise(seed, x, y, 3u, 15, 256, 5, 38) + Q15(0.1);
  water -= 2 * start;
  int32_t mountain = cc_fnoise(seed, x, y, 0u, 15, 256, 5, 77);
  mountain -= 4 * start + (water * 3) / 10;
  if (start > Q15(0.5)) return B_GRASS;
  if (mountain > Q15(0.15)) {
    if (cc_noise(seed, x, y, 6u, 7) > Q15(0.15) && mountain > Q15(0.3)) return B_PATH;         /* caves */
    if (cc_noise(seed, 2 * x, y / 5, 7u, 3) > Q15(0.4)) return B_PATH;                        /* horizontal tunnels */
    if (cc_noise(seed, x / 5, 2 * y, 7u, 3) > Q15(0.4)) return B_PATH;                        /* vertical tunnels */
    if (cc_noise(seed, x, y, 1u, 8) > 0 && u > 55705) return B_COAL;                          /* uniform > 0.85 */
    if (cc_noise(seed, x, y, 2u, 6) > Q15(0.4) && u > 49151) return B_IRON;                   /* > 0.75 */
    if (mountain > Q15(0.18) && u > 65142) return B_DIAMOND;                                  /* > 0.994 */
    if (mountain > Q15(0.3) && cc_noise(seed, x, y, 6u, 5) > Q15(0.35)) return B_LAVA;
    return B_STONE;
  }
  if (water > Q15(0.25) && water <= Q15(0.35) && cc_noise(seed, x, y, 4u, 9) > -Q15(0.2)) return B_SAND;
  if (water > Q15(0.3)) return B_WATER;
  if (cc_noise(seed, x, y, 5u, 7) > 0 && u > 52428) return B_TREE;                            /* > 0.8 */
  return B_GRASS;
}

/* reset_env: world from the env's own seed word; player at the centre, full vitals, empty inventory, no mobs */
void cc_reset_one(uint64_t key, uint32_t e, int32_t *si, float *sf) {
  uint32_t o[2];
  pqn_oracle_env_bits(key, e, ST_WORLD, o);
  const uint32_t seed = o[0];
  for (int r = 0; r < CC_MAP; ++r)
    for (int c = 0; c < CC_MAP; ++c) si[r * CC_MAP + c] = cc_world_cell(seed, r, c);
  int32_t *s = si + CC_S;
  memset(s, 0, sizeof(int32_t) * 111);
  s[0] = 32; s[1] = 32; s[2] = 4;            /* facing down */
  s[3] = 9; s[4] = 9; s[5] = 9; s[6] = 9;
  sf[0] = sf[1] = sf[2] = sf[3] = 0.0f;
}

/* ---- helpers --------------------------------------------------------------------------------------------------- */
static int in_bounds(int r, int c) { return r >= 0 && r < CC_MAP && c >= 0 && c < CC_MAP; }
static int mob_at(const int32_t *s, int r, int c) { /* 1 zombie, 2 cow, 3 skeleton, 0 none (arrows do not block) */
  for (int i = 0; i < CC_NZ; ++i) if (s[20 + 5 * i + 4] && s[20 + 5 * i] == r && s[20 + 5 * i + 1] == c) return 1;
  for (int i = 0; i < CC_NC; ++i) if (s[35 + 4 * i + 3] && s[35 + 4 * i] == r && s[35 + 4 * i + 1] == c) return 2;
  for (int i = 0; i < CC_NS; ++i) if (s[47 + 5 * i + 4] && s[47 + 5 * i] == r && s[47 + 5 * i + 1] == c) return 3;
  return 0;
}
static int near_block(const int32_t *map, int pr, int pc, int block) { /* 3x3 around the player (Crafter nearby(pos, 1)) */
  for (int dr = -1; dr <= 1; ++dr)
    for (int dc = -1; dc <= 1; ++dc)
      if (in_bounds(pr + dr, pc + dc) && map[(pr + dr) * CC_MAP + pc + dc] == block) return 1;
  return 0;
}
static int walkable(int b) { return b == B_GRASS || b == B_SAND || b == B_PATH; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int iabs(int a) { return a < 0 ? -a : a; }
static float uni(uint32_t bits) { return pqn_oracle_bits_to_uniform(bits); }
/* direction (1..4) of one step from (r, c) toward (tr, tc): along the longer axis if `long_axis`, else the shorter
 * one (Crafter objects.toward); ties and zero components fall back to the other axis */
static int toward(int r, int c, int tr, int tc, int long_axis) {
  const int dr = tr - r, dc = tc - c;
  const int vertical_longer = iabs(dr) > iabs(dc);
  int use_vertical = long_axis ? vertical_longer : !vertical_longer;
  if (use_vertical && dr == 0) use_vertical = 0;
  if (!use_vertical && dc == 0) use_vertical = 1;
  if (use_vertical) return dr < 0 ? 3 : 4;
  return dc < 0 ? 1 : 2;
}
static void give(int32_t *s, int ach) { s[109] |= (1 << ach); }

/* step_env for one env; returns reward, sets *done.  s = si + CC_S. */
static float cc_step_one(int32_t *si, float *sf, int32_t action, uint64_t key, uint32_t e, int *done) {
  int32_t *map = si, *s = si + CC_S, *inv = s + 8;
  const int a = s[7] ? A_NOOP : action;                       /* sleeping: the action is replaced by noop */
  const int32_t ach0 = s[109], health0 = s[3];
  uint32_t o[2];
  /* 1. crafting (Crafter data.yaml `make`: nearby table; iron tools also a furnace) */
  {
    const int table = near_block(map, s[0], s[1], B_TABLE), furnace = near_block(map, s[0], s[1], B_FURNACE);
    if (a == A_MAKE_WOOD_PICKAXE && table && inv[I_WOOD] >= 1) { inv[I_WOOD]--; inv[I_WOOD_PICKAXE]++; give(s, ACH_MAKE_WOOD_PICKAXE); }
    if (a == A_MAKE_STONE_PICKAXE && table && inv[I_WOOD] >= 1 && inv[I_STONE] >= 1) { inv[I_WOOD]--; inv[I_STONE]--; inv[I_STONE_PICKAXE]++; give(s, ACH_MAKE_STONE_PICKAXE); }
    if (a == A_MAKE_IRON_PICKAXE && table && furnace && inv[I_WOOD] >= 1 && inv[I_COAL] >= 1 && inv[I_IRON] >= 1) { inv[I_WOOD]--; inv[I_COAL]--; inv[I_IRON]--; inv[I_IRON_PICKAXE]++; give(s, ACH_MAKE_IRON_PICKAXE); }
    if (a == A_MAKE_WOOD_SWORD && table && inv[I_WOOD] >= 1) { inv[I_WOOD]--; inv[I_WOOD_SWORD]++; give(s, ACH_MAKE_WOOD_SWORD); }
    if (a == A_MAKE_STONE_SWORD && table && inv[I_WOOD] >= 1 && inv[I_STONE] >= 1) { inv[I_WOOD]--; inv[I_STONE]--; inv[I_STONE_SWORD]++; give(s, ACH_MAKE_STONE_SWORD); }
    if (a == A_MAKE_IRON_SWORD && table && furnace && inv[I_WOOD] >= 1 && inv[I_COAL] >= 1 && inv[I_IRON] >= 1) { inv[I_WOOD]--; inv[I_COAL]--; inv[I_IRON]--; inv[I_IRON_SWORD]++; give(s, ACH_MAKE_IRON_SWORD); }
  }
  /* 2. interact with the faced cell */
  const int tr = s[0] + DR[s[2]], tc = s[1] + DC[s[2]];
  if (a == A_DO && in_bounds(tr, tc)) {
    const int damage = inv[I_IRON_SWORD] ? 5 : (inv[I_STONE_SWORD] ? 3 : (inv[I_WOOD_SWORD] ? 2 : 1));
    int hit = 0;
    for (int i = 0; i < CC_NZ && !hit; ++i) { int32_t *z = s + 20 + 5 * i;
      if (z[4] && z[0] == tr && z[1] == tc) { hit = 1; z[2] -= damage; if (z[2] <= 0) { z[2] = 0; z[4] = 0; give(s, ACH_DEFEAT_ZOMBIE); } } }
    for (int i = 0; i < CC_NC && !hit; ++i) { int32_t *w = s + 35 + 4 * i;
      if (w[3] && w[0] == tr && w[1] == tc) { hit = 1; w[2] -= damage; if (w[2] <= 0) { w[2] = 0; w[3] = 0; s[4] = imin(s[4] + 6, 9); sf[1] = 0.0f; give(s, ACH_EAT_COW); } } }
    for (int i = 0; i < CC_NS && !hit; ++i) { int32_t *k = s + 47 + 5 * i;
      if (k[4] && k[0] == tr && k[1] == tc) { hit = 1; k[2] -= damage; if (k[2] <= 0) { k[2] = 0; k[4] = 0; give(s, ACH_DEFEAT_SKELETON); } } }
    if (!hit) {
      int32_t *cell = map + tr * CC_MAP + tc;
      switch (*cell) {
        case B_TREE: inv[I_WOOD]++; give(s, ACH_COLLECT_WOOD); break;
        case B_STONE: if (inv[I_WOOD_PICKAXE]) { inv[I_STONE]++; *cell = B_PATH; give(s, ACH_COLLECT_STONE); } break;
        case B_COAL: if (inv[I_WOOD_PICKAXE]) { inv[I_COAL]++; *cell = B_PATH; give(s, ACH_COLLECT_COAL); } break;
        case B_IRON: if (inv[I_STONE_PICKAXE]) { inv[I_IRON]++; *cell = B_PATH; give(s, ACH_COLLECT_IRON); } break;
        case B_DIAMOND: if (inv[I_IRON_PICKAXE]) { inv[I_DIAMOND]++; *cell = B_PATH; give(s, ACH_COLLECT_DIAMOND); } break;
        case B_WATER: s[5] = imin(s[5] + 1, 9); sf[2] = 0.0f; give(s, ACH_COLLECT_DRINK); break;
        case B_GRASS:
          pqn_oracle_env_bits(key, e, ST_SAPLING, o);
          if (uni(o[0]) < 0.1f) { inv[I_SAPLING]++; give(s, ACH_COLLECT_SAPLING); }
          break;
        case B_RIPE:
          *cell = B_PLANT; s[4] = imin(s[4] + 4, 9); sf[1] = 0.0f; give(s, ACH_EAT_PLANT);
          for (int i = 0; i < CC_NP; ++i) { int32_t *p = s + 69 + 4 * i; if (p[3] && p[0] == tr && p[1] == tc) p[2] = 0; }
          break;
        default: break;
      }
    }
  }
  /* 3. placing (Crafter data.yaml `place`) */
  if (a >= A_PLACE_STONE && a <= A_PLACE_PLANT && in_bounds(tr, tc) && !mob_at(s, tr, tc)) {
    int32_t *cell = map + tr * CC_MAP + tc;
    const int b = *cell;
    if (a == A_PLACE_STONE && inv[I_STONE] >= 1 && (walkable(b) || b == B_WATER || b == B_LAVA)) { *cell = B_STONE; inv[I_STONE]--; give(s, ACH_PLACE_STONE); }
    if (a == A_PLACE_TABLE && inv[I_WOOD] >= 1 && walkable(b)) { *cell = B_TABLE; inv[I_WOOD]--; give(s, ACH_PLACE_TABLE); }
    if (a == A_PLACE_FURNACE && inv[I_STONE] >= 1 && walkable(b) && near_block(map, s[0], s[1], B_TABLE)) { *cell = B_FURNACE; inv[I_STONE]--; give(s, ACH_PLACE_FURNACE); }
    if (a == A_PLACE_PLANT && inv[I_SAPLING] >= 1 && b == B_GRASS) {
      for (int i = 0; i < CC_NP; ++i) { int32_t *p = s + 69 + 4 * i;
        if (!p[3]) { p[0] = tr; p[1] = tc; p[2] = 0; p[3] = 1; *cell = B_PLANT; inv[I_SAPLING]--; give(s, ACH_PLACE_PLANT); break; } }
    }
  }
  /* 4. movement: the facing direction always follows the action; lava is enterable (and deadly) */
  if (a >= A_LEFT && a <= A_DOWN) {
    s[2] = a;
    const int nr = s[0] + DR[a], nc = s[1] + DC[a];
    if (in_bounds(nr, nc) && (walkable(map[nr * CC_MAP + nc]) || map[nr * CC_MAP + nc] == B_LAVA) && !mob_at(s, nr, nc)) { s[0] = nr; s[1] = nc; }
  }
  /* 5. mobs */
  for (int i = 0; i < CC_NZ; ++i) { int32_t *z = s + 20 + 5 * i;   /* zombies (Crafter objects.Zombie.update) */
    if (!z[4]) continue;
    pqn_oracle_env_bits(key, e, ST_ZOMBIE + (uint32_t)i, o);
    int dist = imax(iabs(z[0] - s[0]), iabs(z[1] - s[1]));
    int d;
    if (dist <= 8 && uni(o[0]) < 0.9f) d = toward(z[0], z[1], s[0], s[1], (o[1] >> 8) % 10u < 8u);
    else d = 1 + (int)(((uint64_t)o[1] * 4u) >> 32);
    const int nr = z[0] + DR[d], nc = z[1] + DC[d];
    if (in_bounds(nr, nc) && walkable(map[nr * CC_MAP + nc]) && !mob_at(s, nr, nc) && !(nr == s[0] && nc == s[1])) { z[0] = nr; z[1] = nc; }
    dist = imax(iabs(z[0] - s[0]), iabs(z[1] - s[1]));
    if (dist <= 1) {
      if (z[3] > 0) z[3]--;
      else { s[3] -= s[7] ? 7 : 2; z[3] = 5; }
    }
  }
  for (int i = 0; i < CC_NC; ++i) { int32_t *w = s + 35 + 4 * i;   /* cows: a random step half of the time */
    if (!w[3]) continue;
    pqn_oracle_env_bits(key, e, ST_COW + (uint32_t)i, o);
    if (uni(o[0]) > 0.5f) {
      const int d = 1 + (int)(((uint64_t)o[1] * 4u) >> 32);
      const int nr = w[0] + DR[d], nc = w[1] + DC[d];
      if (in_bounds(nr, nc) && walkable(map[nr * CC_MAP + nc]) && !mob_at(s, nr, nc) && !(nr == s[0] && nc == s[1])) { w[0] = nr; w[1] = nc; }
    }
  }
  for (int i = 0; i < CC_NS; ++i) { int32_t *k = s + 47 + 5 * i;   /* skeletons: live on paths, shoot arrows */
    if (!k[4]) continue;
    uint32_t p[2];
    pqn_oracle_env_bits(key, e, ST_SKEL_A + (uint32_t)i, o);
    pqn_oracle_env_bits(key, e, ST_SKEL_B + (uint32_t)i, p);
    k[3] = imax(0, k[3] - 1);
    const int dist = imax(iabs(k[0] - s[0]), iabs(k[1] - s[1]));
    int d = 0;
    if (dist <= 3 && uni(o[0]) < 0.4f) d = toward(k[0], k[1], s[0], s[1], uni(o[1]) < 0.6f);
    else if (dist <= 5 && k[3] == 0 && uni(p[0]) < 0.5f) {       /* shoot along the longer axis toward the player */
      const int ad = toward(k[0], k[1], s[0], s[1], 1);
      const int ar = k[0] + DR[ad], ac = k[1] + DC[ad];
      k[3] = 2;
      if (in_bounds(ar, ac) && map[ar * CC_MAP + ac] == B_PATH && !mob_at(s, ar, ac))
        for (int j = 0; j < CC_NA; ++j) { int32_t *q = s + 57 + 4 * j; if (!q[3]) { q[0] = ar; q[1] = ac; q[2] = ad; q[3] = 1; break; } }
    } else if (dist <= 8 && uni(p[0]) < 0.3f) d = toward(k[0], k[1], s[0], s[1], uni(o[1]) < 0.6f);
    else if (uni(p[1]) < 0.2f) d = 1 + (int)(((uint64_t)o[1] * 4u) >> 32);
    if (d) {
      const int nr = k[0] + DR[d], nc = k[1] + DC[d];
      if (in_bounds(nr, nc) && map[nr * CC_MAP + nc] == B_PATH && !mob_at(s, nr, nc) && !(nr == s[0] && nc == s[1])) { k[0] = nr; k[1] = nc; }
    }
  }
  for (int j = 0; j < CC_NA; ++j) { int32_t *q = s + 57 + 4 * j;   /* arrows fly one cell per step */
    if (!q[3]) continue;
    const int nr = q[0] + DR[q[2]], nc = q[1] + DC[q[2]];
    if (nr == s[0] && nc == s[1]) { s[3] -= 2; q[3] = 0; continue; }
    if (!in_bounds(nr, nc) || mob_at(s, nr, nc)) { q[3] = 0; continue; }
    const int b = map[nr * CC_MAP + nc];
    if (walkable(b) || b == B_WATER || b == B_LAVA) { q[0] = nr; q[1] = nc; }
    else { if (b == B_TABLE || b == B_FURNACE) map[nr * CC_MAP + nc] = B_PATH; q[3] = 0; }
  }
  /* 6. despawn far mobs (Chebyshev distance > 14), then at most one spawn attempt per kind */
  for (int i = 0; i < CC_NZ; ++i) { int32_t *z = s + 20 + 5 * i; if (z[4] && imax(iabs(z[0] - s[0]), iabs(z[1] - s[1])) > 14) z[4] = 0; }
  for (int i = 0; i < CC_NC; ++i) { int32_t *w = s + 35 + 4 * i; if (w[3] && imax(iabs(w[0] - s[0]), iabs(w[1] - s[1])) > 14) w[3] = 0; }
  for (int i = 0; i < CC_NS; ++i) { int32_t *k = s + 47 + 5 * i; if (k[4] && imax(iabs(k[0] - s[0]), iabs(k[1] - s[1])) > 14) k[4] = 0; }
  {
    const float light = cc_light(s[110]);
    const float zchance = 0.02f + 0.1f * ((1.0f - light) * (1.0f - light));
    for (int kind = 0; kind < 3; ++kind) {
      pqn_oracle_env_bits(key, e, ST_SPAWN_COW + (uint32_t)kind, o);
      const float chance = kind == 0 ? 0.1f : (kind == 1 ? zchance : 0.1f);
      if (!(uni(o[0]) < chance)) continue;
      const int r = s[0] + (int)((o[1] & 0xFFFFu) * 19u >> 16) - 9, c = s[1] + (int)((o[1] >> 16) * 19u >> 16) - 9;
      if (!in_bounds(r, c) || mob_at(s, r, c)) continue;
      const int dist = imax(iabs(r - s[0]), iabs(c - s[1])), b = map[r * CC_MAP + c];
      if (kind == 0 && b == B_GRASS && dist >= 4) {
        for (int i = 0; i < CC_NC; ++i) { int32_t *w = s + 35 + 4 * i; if (!w[3]) { w[0] = r; w[1] = c; w[2] = 3; w[3] = 1; break; } }
      } else if (kind == 1 && b == B_GRASS && dist >= 6) {
        for (int i = 0; i < CC_NZ; ++i) { int32_t *z = s + 20 + 5 * i; if (!z[4]) { z[0] = r; z[1] = c; z[2] = 5; z[3] = 0; z[4] = 1; break; } }
      } else if (kind == 2 && b == B_PATH && dist >= 7) {
        for (int i = 0; i < CC_NS; ++i) { int32_t *k = s + 47 + 5 * i; if (!k[4]) { k[0] = r; k[1] = c; k[2] = 3; k[3] = 0; k[4] = 1; break; } }
      }
    }
  }
  /* 7. plants ripen after 300 steps (Crafter objects.Plant: ripe = grown > 300) */
  for (int i = 0; i < CC_NP; ++i) { int32_t *p = s + 69 + 4 * i;
    if (!p[3]) continue;
    int32_t *cell = map + p[0] * CC_MAP + p[1];
    if (*cell != B_PLANT && *cell != B_RIPE) { p[3] = 0; continue; }
    p[2]++;
    if (p[2] > 300) *cell = B_RIPE;
  }
  /* 8. vitals (Crafter objects.Player.update) */
  if (a == A_SLEEP && s[6] < 9) s[7] = 1;
  const int sl = s[7];
  sf[1] += sl ? 0.5f : 1.0f; if (sf[1] > 25.0f) { sf[1] = 0.0f; s[4] = imax(0, s[4] - 1); }
  sf[2] += sl ? 0.5f : 1.0f; if (sf[2] > 20.0f) { sf[2] = 0.0f; s[5] = imax(0, s[5] - 1); }
  if (sl) sf[3] = fminf(sf[3] - 1.0f, 0.0f); else sf[3] += 1.0f;
  if (sf[3] < -10.0f) { sf[3] = 0.0f; s[6] = imin(s[6] + 1, 9); }
  if (sf[3] > 30.0f) { sf[3] = 0.0f; s[6] = imax(0, s[6] - 1); }
  if (s[4] > 0 && s[5] > 0 && (s[6] > 0 || sl)) sf[0] += sl ? 2.0f : 1.0f; else sf[0] -= sl ? 0.5f : 1.0f;
  if (sf[0] > 25.0f) { sf[0] = 0.0f; s[3] = imin(s[3] + 1, 9); }
  if (sf[0] < -15.0f) { sf[0] = 0.0f; s[3] -= 1; }
  if (s[7] && s[6] >= 9) { s[7] = 0; give(s, ACH_WAKE_UP); }            /* rested: wake up */
  /* 9. caps */
  for (int i = 0; i < 12; ++i) inv[i] = imin(inv[i], 9);
  s[3] = imin(s[3], 9);
  /* 10. reward: +1 per newly unlocked achievement, 0.1 per health point gained or lost */
  const uint32_t fresh = (uint32_t)(s[109] & ~ach0);
  float reward = (float)__builtin_popcount(fresh) + 0.1f * (float)(s[3] - health0);
  s[110] += 1;
  *done = (s[110] >= CC_MAX_STEPS) || (s[3] <= 0) || (map[s[0] * CC_MAP + s[1]] == B_LAVA);
  return reward;
}

/* symbolic observation (render_craftax_symbolic): 7 x 9 view centred on the player, 21 channels per cell, then the
 * 22 scalars.  Cells outside the map are OUT_OF_BOUNDS. */
void cc_obs_one(const int32_t *si, float *obs) {
  const int32_t *map = si, *s = si + CC_S;
  memset(obs, 0, sizeof(float) * CC_OBS);
  for (int vr = 0; vr < 7; ++vr)
    for (int vc = 0; vc < 9; ++vc) {
      const int r = s[0] + vr - 3, c = s[1] + vc - 4;
      float *cell = obs + (vr * 9 + vc) * 21;
      const int b = in_bounds(r, c) ? map[r * CC_MAP + c] : B_OOB;
      cell[b] = 1.0f;
      if (in_bounds(r, c)) {
        const int m = mob_at(s, r, c);
        if (m) cell[17 + m - 1] = 1.0f;                                         /* zombie, cow, skeleton */
        for (int j = 0; j < CC_NA; ++j) { const int32_t *q = s + 57 + 4 * j; if (q[3] && q[0] == r && q[1] == c) cell[20] = 1.0f; }
      }
    }
  float *t = obs + 1323;
  for (int i = 0; i < 12; ++i) t[i] = (float)s[8 + i] / 10.0f;
  t[12] = (float)s[3] / 10.0f; t[13] = (float)s[4] / 10.0f; t[14] = (float)s[5] / 10.0f; t[15] = (float)s[6] / 10.0f;
  t[16 + s[2] - 1] = 1.0f;
  t[20] = cc_light(s[110]);
  t[21] = (float)s[7];
}

float cc_step_env(int32_t *si, float *sf, int32_t action, uint64_t key, uint32_t e, int *done) {
  return cc_step_one(si, sf, action, key, e, done);
}
