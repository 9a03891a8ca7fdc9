// pqn_craftax.hip -- Craftax-Classic-Symbolic-v1 for gfx950: the symbolic-grid env of BASELINE.json configs[4]
// (reference call sites purejaxql/pqn_craftax.py:96-99,202-204,433-439; the env itself is the un-vendored third-party
// package craftax>=1.4.2, reference pyproject.toml:50 -- rules restated from the published Crafter / Craftax-Classic
// mechanics, parity unpinned, see oracle/craftax_classic.c for the rule-by-rule statement this file reproduces bit
// for bit).
//
// Layout (SoA of 32-bit words, state[w * n + e], lane = env => every access of a wave is one coalesced row):
//   words 0..1023     the 64 x 64 map, one byte per cell (block id), 4 cells per word, row-major
//   words 1024..1056  packed scalars: player, vitals, inventory, 3 zombies, 3 cows, 2 skeletons, 3 arrows, 10 plants,
//                     achievements, timestep (CcScalars::load / store)
//   + the 5 LogWrapper words of every env kernel
// Kernels per env.step (integer / byte work, HBM- and latency-bound; no matrix work here):
//   cc_step_kernel    lane per env: the transition rule on register-resident scalars; the map is touched through its
//                     SoA words (<= ~150 cell reads, a handful of byte writes per step)
//   cc_reset_kernel   lane per env: which finished envs restart from which reset slot (auto-reset: its own; optimistic
//                     resets: the wrapper's rank rule), scalars of the fresh episode
//   cc_world_kernel   thread per (env, map cell): procedural world of the restarting envs -- a pure function of
//                     (key, slot, cell): fixed-point value noise, so the 4 KB map is regenerated in parallel
//   cc_obs_kernel     workgroup per env: 7 x 9 view -> 1345 f32 (21 one-hot channels per cell + 22 scalars), coalesced
#include <string.h>

#include "pqn_common.h"

#include "pqn_env_rules.h"

namespace cc {
enum { B_INVALID, B_OOB, B_GRASS, B_WATER, B_STONE, B_TREE, B_WOOD, B_PATH, B_COAL, B_IRON, B_DIAMOND, B_TABLE, B_FURNACE,
       B_SAND, B_LAVA, B_PLANT, B_RIPE };
enum { A_NOOP, A_LEFT, A_RIGHT, A_UP, A_DOWN, A_DO, A_SLEEP, A_PLACE_STONE, A_PLACE_TABLE, A_PLACE_FURNACE, A_PLACE_PLANT,
       A_MAKE_WOOD_PICKAXE, A_MAKE_STONE_PICKAXE, A_MAKE_IRON_PICKAXE, A_MAKE_WOOD_SWORD, A_MAKE_STONE_SWORD,
       A_MAKE_IRON_SWORD };
enum { ACH_COLLECT_COAL, ACH_COLLECT_DIAMOND, ACH_COLLECT_DRINK, ACH_COLLECT_IRON, ACH_COLLECT_SAPLING, ACH_COLLECT_STONE,
       ACH_COLLECT_WOOD, ACH_DEFEAT_SKELETON, ACH_DEFEAT_ZOMBIE, ACH_EAT_COW, ACH_EAT_PLANT, ACH_MAKE_IRON_PICKAXE,
       ACH_MAKE_IRON_SWORD, ACH_MAKE_STONE_PICKAXE, ACH_MAKE_STONE_SWORD, ACH_MAKE_WOOD_PICKAXE, ACH_MAKE_WOOD_SWORD,
       ACH_PLACE_FURNACE, ACH_PLACE_PLANT, ACH_PLACE_STONE, ACH_PLACE_TABLE, ACH_WAKE_UP };
enum { I_WOOD, I_STONE, I_COAL, I_IRON, I_DIAMOND, I_SAPLING, I_WOOD_PICKAXE, I_STONE_PICKAXE, I_IRON_PICKAXE, I_WOOD_SWORD,
       I_STONE_SWORD, I_IRON_SWORD };
enum { ST_SAPLING = 10, ST_ZOMBIE = 11, ST_COW = 14, ST_SKEL_A = 17, ST_SKEL_B = 19, ST_SPAWN_COW = 21, ST_WORLD = 40 };
constexpr int MAP = 64, MAP_WORDS = 1024, SCALAR_WORDS = 33, ENV_WORDS = MAP_WORDS + SCALAR_WORDS;
constexpr int OBS = 1345, NUM_ACTIONS = 17, MAX_STEPS = 10000, CANON_SI = 4096 + 111, CANON_SF = 4;
constexpr int NZ = 3, NC = 3, NS = 2, NA = 3, NP = 10;

PQN_HD int dr_of(int d) { return d == 3 ? -1 : (d == 4 ? 1 : 0); }
PQN_HD int dc_of(int d) { return d == 1 ? -1 : (d == 2 ? 1 : 0); }
PQN_HD bool in_bounds(int r, int c) { return (unsigned)r < (unsigned)MAP && (unsigned)c < (unsigned)MAP; }
PQN_HD bool walkable(int b) { return b == B_GRASS || b == B_SAND || b == B_PATH; }
PQN_HD int iabs_(int a) { return a < 0 ? -a : a; }

// daylight: 1 - |cos(pi * (t mod 300) / 300 + 0.3 pi)|^3, with the explicit-f32 cosine shared with the oracle
PQN_HD float light_level(int t) {
  const float progress = (float)(t % 300) / 300.0f + 0.3f;
  float sn, cs;
  pqn_sincos_f32(3.14159265358979323846f * progress, sn, cs);
  const float c = cs < 0.0f ? -cs : cs;
  return 1.0f - c * c * c;
}

// ---- world generation (oracle/craftax_classic.c cc_world_cell, restated) ------------------------------------------
PQN_HD uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
PQN_HD int lattice(uint32_t seed, int ix, int iy, uint32_t layer) {
  const uint32_t h = hash32(seed ^ ((uint32_t)ix * 0x9E3779B1U) ^ ((uint32_t)iy * 0x85EBCA77U) ^ (layer * 0xC2B2AE3DU));
  return (int)(h >> 16) - 32768;
}
PQN_HD int noise(uint32_t seed, int x256, int y256, uint32_t layer, int size) {
  const int span = size * 256;
  const int ix = x256 / span, iy = y256 / span;
  const int fx = ((x256 - ix * span) * 256) / span, fy = ((y256 - iy * span) * 256) / span;
  const int sx = (fx * fx * (768 - 2 * fx)) >> 16, sy = (fy * fy * (768 - 2 * fy)) >> 16;
  const int v00 = lattice(seed, ix, iy, layer), v10 = lattice(seed, ix + 1, iy, layer);
  const int v01 = lattice(seed, ix, iy + 1, layer), v11 = lattice(seed, ix + 1, iy + 1, layer);
  const int a = v00 + (((v10 - v00) * sx) >> 8), b = v01 + (((v11 - v01) * sx) >> 8);
  return a + (((b - a) * sy) >> 8);
}
PQN_HD int fnoise(uint32_t seed, int x256, int y256, uint32_t layer, int s1, int w1, int s2, int w2) {
  const int n1 = noise(seed, x256, y256, layer, s1);
  if (w2 == 0) return n1;
  const int n2 = noise(seed, x256, y256, layer + 16u, s2);
  return (n1 * w1 + n2 * w2) / (w1 + w2);
}
PQN_HD uint32_t isqrt32(uint32_t v) {
  uint32_t r = 0, bit = 1u << 30;
  while (bit > v) bit >>= 2;
  while (bit) {
    if (v >= r + bit) { v -= r + bit; r = (r >> 1) + bit; } else r >>= 1;
    bit >>= 2;
  }
  return r;
}
#define CCQ15(x) ((int)((x) * 32768.0))
PQN_HD int world_cell(uint32_t seed, int r, int c) {
  const int x = c * 256, y = r * 256;
  const uint32_t u = hash32(seed ^ 0xA511E9B3U ^ (uint32_t)(r * 64 + c) * 0x9E3779B1U) >> 16;
  const int dx = c - 32, dy = r - 32;
  const int dist_q8 = (int)isqrt32((uint32_t)(dx * dx + dy * dy) << 16);
  const int sraw = CCQ15(4.0) - dist_q8 * 128 + 2 * noise(seed, x, y, 8u, 3);
  int start = CCQ15(0.5) + sraw / 4;
  start = start < 0 ? 0 : (start > 32767 ? 32767 : start);
  int water = fnoise(seed, x, y, 3u, 15, 256, 5, 38) + CCQ15(0.1);
  water -= 2 * start;
  int mountain = fnoise(seed, x, y, 0u, 15, 256, 5, 77);
  mountain -= 4 * start + (water * 3) / 10;
  if (start > CCQ15(0.5)) return B_GRASS;
  if (mountain > CCQ15(0.15)) {
    if (noise(seed, x, y, 6u, 7) > CCQ15(0.15) && mountain > CCQ15(0.3)) return B_PATH;
    if (noise(seed, 2 * x, y / 5, 7u, 3) > CCQ15(0.4)) return B_PATH;
    if (noise(seed, x / 5, 2 * y, 7u, 3) > CCQ15(0.4)) return B_PATH;
    if (noise(seed, x, y, 1u, 8) > 0 && u > 55705) return B_COAL;
    if (noise(seed, x, y, 2u, 6) > CCQ15(0.4) && u > 49151) return B_IRON;
    if (mountain > CCQ15(0.18) && u > 65142) return B_DIAMOND;
    if (mountain > CCQ15(0.3) && noise(seed, x, y, 6u, 5) > CCQ15(0.35)) return B_LAVA;
    return B_STONE;
  }
  if (water > CCQ15(0.25) && water <= CCQ15(0.35) && noise(seed, x, y, 4u, 9) > -CCQ15(0.2)) return B_SAND;
  if (water > CCQ15(0.3)) return B_WATER;
  if (noise(seed, x, y, 5u, 7) > 0 && u > 52428) return B_TREE;
  return B_GRASS;
}

// ---- per-env scalars in registers -----------------------------------------------------------------------------------
struct Mob { int r, c, health, cd, mask; };      // cd: zombie attack cooldown / skeleton reload / arrow direction
struct Plant { int r, c, age, mask; };
struct Scalars {
  int pr, pc, dir, sleeping, health, food, drink, energy;
  float recover, hunger, thirst, fatigue;
  int inv[12];
  Mob z[NZ], cow[NC], sk[NS], ar[NA];
  Plant pl[NP];
  int ach, timestep;

  static PQN_D uint32_t pack_mob(const Mob &m) {
    return (uint32_t)m.r | ((uint32_t)m.c << 6) | ((uint32_t)(m.health & 15) << 12) | ((uint32_t)(m.cd & 15) << 16) | ((uint32_t)m.mask << 20);
  }
  static PQN_D Mob unpack_mob(uint32_t w) {
    Mob m;
    m.r = w & 63; m.c = (w >> 6) & 63; m.health = (w >> 12) & 15; m.cd = (w >> 16) & 15; m.mask = (w >> 20) & 1;
    return m;
  }
  PQN_D void load(const uint32_t *st, int n, int e) {
    load_from([&](int i) { return st[(size_t)(MAP_WORDS + i) * n + e]; });
  }
  template <class F>
  PQN_D void load_from(F W) {   // W(i) = scalar word i (SCALAR_WORDS of them)
    const uint32_t w0 = W(0), w2 = W(2);
    pr = w0 & 255; pc = (w0 >> 8) & 255; dir = (w0 >> 16) & 255; sleeping = (w0 >> 24) & 1;
    health = (int)W(1);
    food = w2 & 255; drink = (w2 >> 8) & 255; energy = (w2 >> 16) & 255;
    recover = __uint_as_float(W(3)); hunger = __uint_as_float(W(4)); thirst = __uint_as_float(W(5)); fatigue = __uint_as_float(W(6));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint32_t w = W(7 + k);
#pragma unroll
      for (int j = 0; j < 4; ++j) inv[4 * k + j] = (w >> (8 * j)) & 255;
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) z[i] = unpack_mob(W(10 + i));
#pragma unroll
    for (int i = 0; i < NC; ++i) cow[i] = unpack_mob(W(13 + i));
#pragma unroll
    for (int i = 0; i < NS; ++i) sk[i] = unpack_mob(W(16 + i));
#pragma unroll
    for (int i = 0; i < NA; ++i) ar[i] = unpack_mob(W(18 + i));
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const uint32_t w = W(21 + i);
      pl[i].r = w & 63; pl[i].c = (w >> 6) & 63; pl[i].age = (w >> 12) & 16383; pl[i].mask = (w >> 26) & 1;
    }
    ach = (int)W(31);
    timestep = (int)W(32);
  }
  PQN_D void store(uint32_t *st, int n, int e) const {
    auto W = [&](int i, uint32_t v) { st[(size_t)(MAP_WORDS + i) * n + e] = v; };
    W(0, (uint32_t)pr | ((uint32_t)pc << 8) | ((uint32_t)dir << 16) | ((uint32_t)sleeping << 24));
    W(1, (uint32_t)health);
    W(2, (uint32_t)food | ((uint32_t)drink << 8) | ((uint32_t)energy << 16));
    W(3, __float_as_uint(recover)); W(4, __float_as_uint(hunger)); W(5, __float_as_uint(thirst)); W(6, __float_as_uint(fatigue));
#pragma unroll
    for (int k = 0; k < 3; ++k)
      W(7 + k, (uint32_t)inv[4 * k] | ((uint32_t)inv[4 * k + 1] << 8) | ((uint32_t)inv[4 * k + 2] << 16) | ((uint32_t)inv[4 * k + 3] << 24));
#pragma unroll
    for (int i = 0; i < NZ; ++i) W(10 + i, pack_mob(z[i]));
#pragma unroll
    for (int i = 0; i < NC; ++i) W(13 + i, pack_mob(cow[i]));
#pragma unroll
    for (int i = 0; i < NS; ++i) W(16 + i, pack_mob(sk[i]));
#pragma unroll
    for (int i = 0; i < NA; ++i) W(18 + i, pack_mob(ar[i]));
#pragma unroll
    for (int i = 0; i < NP; ++i)
      W(21 + i, (uint32_t)pl[i].r | ((uint32_t)pl[i].c << 6) | ((uint32_t)(pl[i].age & 16383) << 12) | ((uint32_t)pl[i].mask << 26));
    W(31, (uint32_t)ach);
    W(32, (uint32_t)timestep);
  }
  PQN_D void fresh() {   // reset_env: centre of the map, facing down, full vitals, nothing else
    pr = 32; pc = 32; dir = 4; sleeping = 0; health = 9; food = 9; drink = 9; energy = 9;
    recover = hunger = thirst = fatigue = 0.0f;
#pragma unroll
    for (int i = 0; i < 12; ++i) inv[i] = 0;
    const Mob none = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NZ; ++i) z[i] = none;
#pragma unroll
    for (int i = 0; i < NC; ++i) cow[i] = none;
#pragma unroll
    for (int i = 0; i < NS; ++i) sk[i] = none;
#pragma unroll
    for (int i = 0; i < NA; ++i) ar[i] = none;
#pragma unroll
    for (int i = 0; i < NP; ++i) pl[i] = Plant{0, 0, 0, 0};
    ach = 0;
    timestep = 0;
  }
  PQN_D int mob_at(int r, int c) const {   // 1 zombie, 2 cow, 3 skeleton
#pragma unroll
    for (int i = 0; i < NZ; ++i) if (z[i].mask && z[i].r == r && z[i].c == c) return 1;
#pragma unroll
    for (int i = 0; i < NC; ++i) if (cow[i].mask && cow[i].r == r && cow[i].c == c) return 2;
#pragma unroll
    for (int i = 0; i < NS; ++i) if (sk[i].mask && sk[i].r == r && sk[i].c == c) return 3;
    return 0;
  }
};

// map access through the SoA words of env e
struct MapRef {
  uint32_t *st;
  int n, e;
  PQN_D int get(int r, int c) const {
    const int cell = r * MAP + c;
    return (st[(size_t)(cell >> 2) * n + e] >> (8 * (cell & 3))) & 255;
  }
  PQN_D void set(int r, int c, int b) const {
    const int cell = r * MAP + c, sh = 8 * (cell & 3);
    uint32_t *w = st + (size_t)(cell >> 2) * n + e;
    *w = (*w & ~(255u << sh)) | ((uint32_t)b << sh);
  }
  PQN_D bool near(int pr, int pc, int block) const {
    for (int dr = -1; dr <= 1; ++dr)
      for (int dc = -1; dc <= 1; ++dc)
        if (in_bounds(pr + dr, pc + dc) && get(pr + dr, pc + dc) == block) return true;
    return false;
  }
};

// The 3 x 3 cells around the player's position at the start of a step, loaded as ONE batch of nine independent reads and
// kept coherent with the step's writes: every cell the player's own action reads or writes lies in it.  (The rule used to
// walk them one dependent load at a time -- near(table), near(furnace), the faced cell, near(table) again, the cell moved
// to: up to ~30 L2 round trips at the head of a step whose whole kernel is latency.)
struct Hood {
  int r0, c0;
  int v[9];   // -1 = out of bounds
  PQN_D void load(const MapRef &m, int pr, int pc) {
    r0 = pr; c0 = pc;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int r = pr + k / 3 - 1, c = pc + k % 3 - 1;
      const int b = m.get(min(max(r, 0), MAP - 1), min(max(c, 0), MAP - 1));   // unconditional: all nine in flight
      v[k] = in_bounds(r, c) ? b : -1;
    }
  }
  PQN_D bool has(int r, int c) const { return iabs_(r - r0) <= 1 && iabs_(c - c0) <= 1; }
  PQN_D int at(int r, int c) const {
    const int k = (r - r0 + 1) * 3 + (c - c0 + 1);
    int b = v[0];
#pragma unroll
    for (int j = 1; j < 9; ++j) b = k == j ? v[j] : b;
    return b;
  }
  PQN_D void put(int r, int c, int b) {
    const int k = (r - r0 + 1) * 3 + (c - c0 + 1);
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = k == j ? b : v[j];
  }
  PQN_D bool near(int block) const {
    bool f = false;
#pragma unroll
    for (int j = 0; j < 9; ++j) f = f || v[j] == block;
    return f;
  }
};
// the map as the rule sees it: the neighbourhood from registers, everything else from memory; writes go to both
struct MapView {
  const MapRef &m;
  Hood &h;
  PQN_D int get(int r, int c) const { return h.has(r, c) ? h.at(r, c) : m.get(r, c); }
  PQN_D void set(int r, int c, int b) const {
    m.set(r, c, b);
    if (h.has(r, c)) h.put(r, c, b);
  }
};

PQN_D int toward(int r, int c, int tr, int tc, bool long_axis) {
  const int dr = tr - r, dc = tc - c;
  const bool vertical_longer = iabs_(dr) > iabs_(dc);
  bool use_vertical = long_axis ? vertical_longer : !vertical_longer;
  if (use_vertical && dr == 0) use_vertical = false;
  if (!use_vertical && dc == 0) use_vertical = true;
  if (use_vertical) return dr < 0 ? 3 : 4;
  return dc < 0 ? 1 : 2;
}

// the transition rule (oracle/craftax_classic.c cc_step_one, rule for rule)
PQN_D float step(Scalars &s, const MapRef &mref, int action, uint64_t key, uint32_t e, int &done) {
  const int a = s.sleeping ? (int)A_NOOP : action;
  const int ach0 = s.ach, health0 = s.health;
  uint32_t o0, o1;
  auto give = [&](int k) { s.ach |= (1 << k); };
  int *inv = s.inv;
  Hood hood;
  hood.load(mref, s.pr, s.pc);
  const MapView map{mref, hood};
  {  // 1. crafting
    const bool table = hood.near(B_TABLE), furnace = hood.near(B_FURNACE);
    if (a == A_MAKE_WOOD_PICKAXE && table && inv[I_WOOD] >= 1) { inv[I_WOOD]--; inv[I_WOOD_PICKAXE]++; give(ACH_MAKE_WOOD_PICKAXE); }
    if (a == A_MAKE_STONE_PICKAXE && table && inv[I_WOOD] >= 1 && inv[I_STONE] >= 1) { inv[I_WOOD]--; inv[I_STONE]--; inv[I_STONE_PICKAXE]++; give(ACH_MAKE_STONE_PICKAXE); }
    if (a == A_MAKE_IRON_PICKAXE && table && furnace && inv[I_WOOD] >= 1 && inv[I_COAL] >= 1 && inv[I_IRON] >= 1) { inv[I_WOOD]--; inv[I_COAL]--; inv[I_IRON]--; inv[I_IRON_PICKAXE]++; give(ACH_MAKE_IRON_PICKAXE); }
    if (a == A_MAKE_WOOD_SWORD && table && inv[I_WOOD] >= 1) { inv[I_WOOD]--; inv[I_WOOD_SWORD]++; give(ACH_MAKE_WOOD_SWORD); }
    if (a == A_MAKE_STONE_SWORD && table && inv[I_WOOD] >= 1 && inv[I_STONE] >= 1) { inv[I_WOOD]--; inv[I_STONE]--; inv[I_STONE_SWORD]++; give(ACH_MAKE_STONE_SWORD); }
    if (a == A_MAKE_IRON_SWORD && table && furnace && inv[I_WOOD] >= 1 && inv[I_COAL] >= 1 && inv[I_IRON] >= 1) { inv[I_WOOD]--; inv[I_COAL]--; inv[I_IRON]--; inv[I_IRON_SWORD]++; give(ACH_MAKE_IRON_SWORD); }
  }
  // 2. interact with the faced cell
  const int tr = s.pr + dr_of(s.dir), tc = s.pc + dc_of(s.dir);
  if (a == A_DO && in_bounds(tr, tc)) {
    const int damage = inv[I_IRON_SWORD] ? 5 : (inv[I_STONE_SWORD] ? 3 : (inv[I_WOOD_SWORD] ? 2 : 1));
    bool hit = false;
#pragma unroll
    for (int i = 0; i < NZ; ++i)
      if (!hit && s.z[i].mask && s.z[i].r == tr && s.z[i].c == tc) { hit = true; s.z[i].health -= damage; if (s.z[i].health <= 0) { s.z[i].mask = 0; s.z[i].health = 0; give(ACH_DEFEAT_ZOMBIE); } }
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (!hit && s.cow[i].mask && s.cow[i].r == tr && s.cow[i].c == tc) { hit = true; s.cow[i].health -= damage; if (s.cow[i].health <= 0) { s.cow[i].mask = 0; s.cow[i].health = 0; s.food = min(s.food + 6, 9); s.hunger = 0.0f; give(ACH_EAT_COW); } }
#pragma unroll
    for (int i = 0; i < NS; ++i)
      if (!hit && s.sk[i].mask && s.sk[i].r == tr && s.sk[i].c == tc) { hit = true; s.sk[i].health -= damage; if (s.sk[i].health <= 0) { s.sk[i].mask = 0; s.sk[i].health = 0; give(ACH_DEFEAT_SKELETON); } }
    if (!hit) {
      const int b = map.get(tr, tc);
      if (b == B_TREE) { inv[I_WOOD]++; give(ACH_COLLECT_WOOD); }
      else if (b == B_STONE) { if (inv[I_WOOD_PICKAXE]) { inv[I_STONE]++; map.set(tr, tc, B_PATH); give(ACH_COLLECT_STONE); } }
      else if (b == B_COAL) { if (inv[I_WOOD_PICKAXE]) { inv[I_COAL]++; map.set(tr, tc, B_PATH); give(ACH_COLLECT_COAL); } }
      else if (b == B_IRON) { if (inv[I_STONE_PICKAXE]) { inv[I_IRON]++; map.set(tr, tc, B_PATH); give(ACH_COLLECT_IRON); } }
      else if (b == B_DIAMOND) { if (inv[I_IRON_PICKAXE]) { inv[I_DIAMOND]++; map.set(tr, tc, B_PATH); give(ACH_COLLECT_DIAMOND); } }
      else if (b == B_WATER) { s.drink = min(s.drink + 1, 9); s.thirst = 0.0f; give(ACH_COLLECT_DRINK); }
      else if (b == B_GRASS) {
        pqn_bits(key, e, ST_SAPLING, o0, o1);
        if (pqn_uniform(o0) < 0.1f) { inv[I_SAPLING]++; give(ACH_COLLECT_SAPLING); }
      } else if (b == B_RIPE) {
        map.set(tr, tc, B_PLANT); s.food = min(s.food + 4, 9); s.hunger = 0.0f; give(ACH_EAT_PLANT);
#pragma unroll
        for (int i = 0; i < NP; ++i) if (s.pl[i].mask && s.pl[i].r == tr && s.pl[i].c == tc) s.pl[i].age = 0;
      }
    }
  }
  // 3. placing
  if (a >= A_PLACE_STONE && a <= A_PLACE_PLANT && in_bounds(tr, tc) && !s.mob_at(tr, tc)) {
    const int b = map.get(tr, tc);
    if (a == A_PLACE_STONE && inv[I_STONE] >= 1 && (walkable(b) || b == B_WATER || b == B_LAVA)) { map.set(tr, tc, B_STONE); inv[I_STONE]--; give(ACH_PLACE_STONE); }
    if (a == A_PLACE_TABLE && inv[I_WOOD] >= 1 && walkable(b)) { map.set(tr, tc, B_TABLE); inv[I_WOOD]--; give(ACH_PLACE_TABLE); }
    if (a == A_PLACE_FURNACE && inv[I_STONE] >= 1 && walkable(b) && hood.near(B_TABLE)) { map.set(tr, tc, B_FURNACE); inv[I_STONE]--; give(ACH_PLACE_FURNACE); }
    if (a == A_PLACE_PLANT && inv[I_SAPLING] >= 1 && b == B_GRASS) {
      bool placed = false;
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (!placed && !s.pl[i].mask) { s.pl[i] = Plant{tr, tc, 0, 1}; map.set(tr, tc, B_PLANT); inv[I_SAPLING]--; give(ACH_PLACE_PLANT); placed = true; }
    }
  }
  // 4. movement
  if (a >= A_LEFT && a <= A_DOWN) {
    s.dir = a;
    const int nr = s.pr + dr_of(a), nc = s.pc + dc_of(a);
    if (in_bounds(nr, nc)) {
      const int b = map.get(nr, nc);
      if ((walkable(b) || b == B_LAVA) && !s.mob_at(nr, nc)) { s.pr = nr; s.pc = nc; }
    }
  }
  // 5. mobs.  What a mob wants (direction, target cell) depends on its own position, the player's and its own random draws
  // only, and no mob writes the map: every wish is decided first, the target cells are fetched as one batch, then the
  // moves are applied in the oracle's order (zombies, cows, skeletons; mob_at sees the earlier moves).
  int zr[NZ], zc[NZ], zb[NZ], cr[NC], cc_[NC], cb[NC], kr[NS], kc[NS], kb[NS], kdir[NS];
  bool kshoot[NS];
#pragma unroll
  for (int i = 0; i < NZ; ++i) {
    const Mob &z = s.z[i];
    zr[i] = zc[i] = 0; zb[i] = -1;
    if (!z.mask) continue;
    pqn_bits(key, e, ST_ZOMBIE + (uint32_t)i, o0, o1);
    const int dist = max(iabs_(z.r - s.pr), iabs_(z.c - s.pc));
    int d;
    if (dist <= 8 && pqn_uniform(o0) < 0.9f) d = toward(z.r, z.c, s.pr, s.pc, (o1 >> 8) % 10u < 8u);
    else d = 1 + (int)pqn_randint(o1, 4u);
    zr[i] = z.r + dr_of(d); zc[i] = z.c + dc_of(d);
    if (in_bounds(zr[i], zc[i])) zb[i] = map.get(zr[i], zc[i]);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const Mob &w = s.cow[i];
    cr[i] = cc_[i] = 0; cb[i] = -1;
    if (!w.mask) continue;
    pqn_bits(key, e, ST_COW + (uint32_t)i, o0, o1);
    if (pqn_uniform(o0) > 0.5f) {
      const int d = 1 + (int)pqn_randint(o1, 4u);
      cr[i] = w.r + dr_of(d); cc_[i] = w.c + dc_of(d);
      if (in_bounds(cr[i], cc_[i])) cb[i] = map.get(cr[i], cc_[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    Mob &k = s.sk[i];
    kr[i] = kc[i] = 0; kb[i] = -1; kdir[i] = 0; kshoot[i] = false;
    if (!k.mask) continue;
    uint32_t p0, p1;
    pqn_bits(key, e, ST_SKEL_A + (uint32_t)i, o0, o1);
    pqn_bits(key, e, ST_SKEL_B + (uint32_t)i, p0, p1);
    k.cd = max(0, k.cd - 1);
    const int dist = max(iabs_(k.r - s.pr), iabs_(k.c - s.pc));
    int d = 0;
    if (dist <= 3 && pqn_uniform(o0) < 0.4f) d = toward(k.r, k.c, s.pr, s.pc, pqn_uniform(o1) < 0.6f);
    else if (dist <= 5 && k.cd == 0 && pqn_uniform(p0) < 0.5f) {
      kdir[i] = toward(k.r, k.c, s.pr, s.pc, true);   // the arrow's direction
      kshoot[i] = true;
      k.cd = 2;
    } else if (dist <= 8 && pqn_uniform(p0) < 0.3f) d = toward(k.r, k.c, s.pr, s.pc, pqn_uniform(o1) < 0.6f);
    else if (pqn_uniform(p1) < 0.2f) d = 1 + (int)pqn_randint(o1, 4u);
    if (kshoot[i]) d = kdir[i];
    else kdir[i] = d;
    if (d) {   // the cell the arrow would start in / the skeleton would step to
      kr[i] = k.r + dr_of(d); kc[i] = k.c + dc_of(d);
      if (in_bounds(kr[i], kc[i])) kb[i] = map.get(kr[i], kc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NZ; ++i) {
    Mob &z = s.z[i];
    if (!z.mask) continue;
    const int nr = zr[i], nc = zc[i];
    if (in_bounds(nr, nc) && walkable(zb[i]) && !s.mob_at(nr, nc) && !(nr == s.pr && nc == s.pc)) { z.r = nr; z.c = nc; }
    const int dist = max(iabs_(z.r - s.pr), iabs_(z.c - s.pc));
    if (dist <= 1) {
      if (z.cd > 0) z.cd--;
      else { s.health -= s.sleeping ? 7 : 2; z.cd = 5; }
    }
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    Mob &w = s.cow[i];
    if (!w.mask || cb[i] < 0) continue;   // no wish this step, or out of bounds
    const int nr = cr[i], nc = cc_[i];
    if (walkable(cb[i]) && !s.mob_at(nr, nc) && !(nr == s.pr && nc == s.pc)) { w.r = nr; w.c = nc; }
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    Mob &k = s.sk[i];
    if (!k.mask || kdir[i] == 0) continue;
    const int nr = kr[i], nc = kc[i];
    if (kshoot[i]) {
      if (in_bounds(nr, nc) && kb[i] == B_PATH && !s.mob_at(nr, nc)) {
        bool shot = false;
#pragma unroll
        for (int j = 0; j < NA; ++j)
          if (!shot && !s.ar[j].mask) { s.ar[j] = Mob{nr, nc, 0, kdir[i], 1}; shot = true; }
      }
    } else if (in_bounds(nr, nc) && kb[i] == B_PATH && !s.mob_at(nr, nc) && !(nr == s.pr && nc == s.pc)) { k.r = nr; k.c = nc; }
  }
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    Mob &q = s.ar[j];     // q.cd = direction
    if (!q.mask) continue;
    const int nr = q.r + dr_of(q.cd), nc = q.c + dc_of(q.cd);
    if (nr == s.pr && nc == s.pc) { s.health -= 2; q.mask = 0; continue; }
    if (!in_bounds(nr, nc) || s.mob_at(nr, nc)) { q.mask = 0; continue; }
    const int b = map.get(nr, nc);
    if (walkable(b) || b == B_WATER || b == B_LAVA) { q.r = nr; q.c = nc; }
    else { if (b == B_TABLE || b == B_FURNACE) map.set(nr, nc, B_PATH); q.mask = 0; }
  }
  // 6. despawn / spawn
#pragma unroll
  for (int i = 0; i < NZ; ++i) if (s.z[i].mask && max(iabs_(s.z[i].r - s.pr), iabs_(s.z[i].c - s.pc)) > 14) s.z[i].mask = 0;
#pragma unroll
  for (int i = 0; i < NC; ++i) if (s.cow[i].mask && max(iabs_(s.cow[i].r - s.pr), iabs_(s.cow[i].c - s.pc)) > 14) s.cow[i].mask = 0;
#pragma unroll
  for (int i = 0; i < NS; ++i) if (s.sk[i].mask && max(iabs_(s.sk[i].r - s.pr), iabs_(s.sk[i].c - s.pc)) > 14) s.sk[i].mask = 0;
  {
    const float light = light_level(s.timestep);
    const float zchance = 0.02f + 0.1f * ((1.0f - light) * (1.0f - light));
    int sr[3], sc[3], sb[3];   // candidate cell of each kind and its block (-1: no spawn attempt / out of bounds), one batch
#pragma unroll
    for (int kind = 0; kind < 3; ++kind) {
      pqn_bits(key, e, ST_SPAWN_COW + (uint32_t)kind, o0, o1);
      const float chance = kind == 0 ? 0.1f : (kind == 1 ? zchance : 0.1f);
      sr[kind] = s.pr + (int)(((o1 & 0xFFFFu) * 19u) >> 16) - 9;
      sc[kind] = s.pc + (int)(((o1 >> 16) * 19u) >> 16) - 9;
      sb[kind] = -1;
      if (pqn_uniform(o0) < chance && in_bounds(sr[kind], sc[kind])) sb[kind] = map.get(sr[kind], sc[kind]);
    }
#pragma unroll
    for (int kind = 0; kind < 3; ++kind) {
      const int r = sr[kind], c = sc[kind], b = sb[kind];
      if (b < 0 || s.mob_at(r, c)) continue;   // mob_at sees the mobs the earlier kinds just spawned
      const int dist = max(iabs_(r - s.pr), iabs_(c - s.pc));
      bool placed = false;
      if (kind == 0 && b == B_GRASS && dist >= 4) {
#pragma unroll
        for (int i = 0; i < NC; ++i) if (!placed && !s.cow[i].mask) { s.cow[i] = Mob{r, c, 3, 0, 1}; placed = true; }
      } else if (kind == 1 && b == B_GRASS && dist >= 6) {
#pragma unroll
        for (int i = 0; i < NZ; ++i) if (!placed && !s.z[i].mask) { s.z[i] = Mob{r, c, 5, 0, 1}; placed = true; }
      } else if (kind == 2 && b == B_PATH && dist >= 7) {
#pragma unroll
        for (int i = 0; i < NS; ++i) if (!placed && !s.sk[i].mask) { s.sk[i] = Mob{r, c, 3, 0, 1}; placed = true; }
      }
    }
  }
  // 7. plants
  int pb[NP];   // one batch (a plant's own write below touches its own cell only)
#pragma unroll
  for (int i = 0; i < NP; ++i) pb[i] = s.pl[i].mask ? map.get(s.pl[i].r, s.pl[i].c) : -1;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    Plant &p = s.pl[i];
    if (!p.mask) continue;
    const int b = pb[i];
    if (b != B_PLANT && b != B_RIPE) { p.mask = 0; continue; }
    p.age++;
    if (p.age > 300) map.set(p.r, p.c, B_RIPE);
  }
  // 8. vitals
  if (a == A_SLEEP && s.energy < 9) s.sleeping = 1;
  const bool sl = s.sleeping != 0;
  s.hunger += sl ? 0.5f : 1.0f; if (s.hunger > 25.0f) { s.hunger = 0.0f; s.food = max(0, s.food - 1); }
  s.thirst += sl ? 0.5f : 1.0f; if (s.thirst > 20.0f) { s.thirst = 0.0f; s.drink = max(0, s.drink - 1); }
  if (sl) s.fatigue = fminf(s.fatigue - 1.0f, 0.0f); else s.fatigue += 1.0f;
  if (s.fatigue < -10.0f) { s.fatigue = 0.0f; s.energy = min(s.energy + 1, 9); }
  if (s.fatigue > 30.0f) { s.fatigue = 0.0f; s.energy = max(0, s.energy - 1); }
  if (s.food > 0 && s.drink > 0 && (s.energy > 0 || sl)) s.recover += sl ? 2.0f : 1.0f; else s.recover -= sl ? 0.5f : 1.0f;
  if (s.recover > 25.0f) { s.recover = 0.0f; s.health = min(s.health + 1, 9); }
  if (s.recover < -15.0f) { s.recover = 0.0f; s.health -= 1; }
  if (s.sleeping && s.energy >= 9) { s.sleeping = 0; give(ACH_WAKE_UP); }
  // 9. caps
#pragma unroll
  for (int i = 0; i < 12; ++i) inv[i] = min(inv[i], 9);
  s.health = min(s.health, 9);
  // 10. reward
  const uint32_t fresh = (uint32_t)(s.ach & ~ach0);
  const float reward = (float)__popc(fresh) + 0.1f * (float)(s.health - health0);
  s.timestep += 1;
  done = (s.timestep >= MAX_STEPS) || (s.health <= 0) || (map.get(s.pr, s.pc) == B_LAVA);
  return reward;
}
}  // namespace cc

// ======================================================================================================================
// kernels
// ======================================================================================================================
// lane per env: step_env + LogWrapper.  No reset here: `done` is written and cc_reset_kernel decides what restarts.
__global__ __launch_bounds__(256) void cc_step_kernel(int n, uint64_t key, const uint64_t *__restrict__ key_dev, float rscale,
                                                      uint32_t *state, const int32_t *__restrict__ action, pqn_step_out_t out,
                                                      uint64_t *__restrict__ opt_keys) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  if (key_dev) key = *key_dev;
  cc::Scalars s;
  s.load(state, n, e);
  LogRec log;
  log.load(state, n, e, cc::ENV_WORDS);
  const cc::MapRef map = {state, n, e};
  int done = 0;
  const float reward = cc::step(s, map, action[e], key, (uint32_t)e, done);
  log.step(reward, done);
  s.store(state, n, e);
  log.store(state, n, e, cc::ENV_WORDS);
  out.reward[e] = reward * rscale;
  out.done[e] = (uint8_t)done;
  if (out.discount) out.discount[e] = done ? 0.0f : 1.0f;
  if (out.returned_episode_returns) out.returned_episode_returns[e] = log.ret_ret;
  if (out.returned_episode_lengths) out.returned_episode_lengths[e] = log.ret_len;
  if (out.timestep) out.timestep[e] = log.timestep;
  if (out.achievements) out.achievements[e] = done ? (uint32_t)s.ach : 0u;
  if (opt_keys) {   // optimistic resets: the sort key that ranks the finished envs (see pqn_env.hip opt_choice_key)
    uint64_t k = ~(uint64_t)0;
    if (done) {
      uint32_t o0, o1;
      pqn_bits(pqn_fold(key, 2u), (uint32_t)e, 0u, o0, o1);
      k = ((uint64_t)(o0 >> 1) << 32) | (uint32_t)e;
    }
    opt_keys[e] = k;
  }
}

// lane per env: which envs restart, from which reset slot, and the scalars of the fresh episode.
//   mode 0  pqn_env_reset: every env, slot = e
//   mode 1  auto-reset (gymnax semantics): finished envs, slot = e, the LogWrapper record keeps running
//   mode 2  optimistic resets: finished envs, slot by rank (utils/craftax_wrappers.py:118-135), LogWrapper restarts
__global__ __launch_bounds__(256) void cc_reset_kernel(int n, int mode, int reset_ratio, const uint8_t *__restrict__ done,
                                                       const uint64_t *__restrict__ opt_keys, uint32_t *state,
                                                       int32_t *__restrict__ slots) {
  __shared__ int s_list[256], s_rank[256], s_nd;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const bool mine_done = e < n && mode != 0 && done[e];
  int rank = 0;
  if (mode == 2) {   // rank of a finished env = number of envs with a smaller sort key (the keys of unfinished envs sort last)
    // The finished envs of this workgroup are few: for each, ALL 256 threads count a strided share of the n keys (ballot +
    // popcount per wave, one LDS atomic per wave: integer sums, order-free).  One lane per finished env walking all n keys
    // was 1024 x (LDS read, 64-bit compare, add) in a single wave: 20 us of a 24-us launch at 1024 envs.
    if (threadIdx.x == 0) s_nd = 0;
    s_rank[threadIdx.x] = 0;
    __syncthreads();
    if (mine_done) s_list[atomicAdd(&s_nd, 1)] = threadIdx.x;   // list order does not matter
    __syncthreads();
    const int nd = s_nd;
    for (int d = 0; d < nd; ++d) {
      const int t = s_list[d];
      const uint64_t key = opt_keys[blockIdx.x * 256 + t];
      int cnt = 0;
      for (int j0 = 0; j0 < n; j0 += 256 * 4) {
        uint64_t k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = j0 + 256 * u + threadIdx.x; k[u] = j < n ? opt_keys[j] : ~0ull; }
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt += __popcll(__ballot(k[u] < key));
      }
      if ((threadIdx.x & 63) == 0) atomicAdd(&s_rank[t], cnt);
    }
    __syncthreads();
    rank = s_rank[threadIdx.x];
  }
  if (e >= n) return;
  int slot = -1;
  if (mode == 0) slot = e;
  else if (mine_done) slot = mode == 2 ? (rank < n / reset_ratio ? rank : e / reset_ratio) : e;
  slots[e] = slot;
  if (slot < 0) return;
  cc::Scalars s;
  s.fresh();
  s.store(state, n, e);
  if (mode != 1) {
    LogRec log;
    log.zero();
    log.store(state, n, e, cc::ENV_WORDS);
  }
}

// the world of a restarting env: thread per CELL, grid (env, 16 blocks of 256 cells); the four lanes of a map word pack
// their bytes with two lane exchanges.  Few envs restart per step, so the parallelism has to come from the cells of one
// world: a lane-per-env form left one lane of a wave generating 4096 cells in sequence (121 us per step at 1024 envs), 16
// cells per thread still 38 us on average (world_cell is ~1000 integer instructions); the other workgroups leave at once.
__global__ __launch_bounds__(256) void cc_world_kernel(int n, uint64_t key, const uint64_t *__restrict__ key_dev, int fold_reset,
                                                       const int32_t *__restrict__ slots, uint32_t *__restrict__ state) {
  const int e = blockIdx.x;
  const int slot = slots[e];
  if (slot < 0) return;
  if (key_dev) key = *key_dev;
  if (fold_reset) key = pqn_fold(key, 1u);   // optimistic resets draw their worlds from fold_in(key, 1)
  uint32_t o0, o1;
  pqn_bits(key, (uint32_t)slot, cc::ST_WORLD, o0, o1);
  const int cell = blockIdx.y * 256 + threadIdx.x;
  uint32_t w = (uint32_t)cc::world_cell(o0, cell >> 6, cell & 63) << (8 * (cell & 3));
  w |= __shfl_xor(w, 1);
  w |= __shfl_xor(w, 2);
  if ((cell & 3) == 0) state[(size_t)(cell >> 2) * n + e] = w;
}

// ---- the symbolic observation: 7 x 9 view x 21 one-hot channels + 22 scalars = 1345 f32 per env --------------------------
// view cell k of the env whose scalar words are w: block id | mob channel bits << 8
PQN_D int cc_obs_cell(const uint32_t *w, const uint32_t *__restrict__ state, int n, int e, int k) {
  cc::Scalars s;
  s.load_from([&](int i) { return w[i]; });
  const int vr = k / 9, vc = k - 9 * vr;
  const int r = s.pr + vr - 3, c = s.pc + vc - 4;
  int v = cc::B_OOB;
  if (cc::in_bounds(r, c)) {
    const int cell = r * cc::MAP + c;
    v = (state[(size_t)(cell >> 2) * n + e] >> (8 * (cell & 3))) & 255;
    const int m = s.mob_at(r, c);
    if (m) v |= 1 << (8 + m - 1);
#pragma unroll
    for (int j = 0; j < cc::NA; ++j) if (s.ar[j].mask && s.ar[j].r == r && s.ar[j].c == c) v |= 1 << 11;
  }
  return v;
}
// scalar k of the 22 behind the view, straight from the packed words (Scalars::load_from's layout; indexing the unpacked
// inventory with a run-time k would put the struct into scratch memory)
PQN_D float cc_obs_tail(const uint32_t *w, int k) {
  const uint32_t w0 = w[0], w2 = w[2];
  if (k < 12) return (float)((w[7 + (k >> 2)] >> (8 * (k & 3))) & 255u) / 10.0f;   // inventory
  if (k == 12) return (float)(int)w[1] / 10.0f;                                   // health
  if (k == 13) return (float)(w2 & 255u) / 10.0f;                                 // food
  if (k == 14) return (float)((w2 >> 8) & 255u) / 10.0f;                          // drink
  if (k == 15) return (float)((w2 >> 16) & 255u) / 10.0f;                         // energy
  if (k < 20) return (k - 16 == (int)((w0 >> 16) & 255u) - 1) ? 1.0f : 0.0f;      // direction one-hot
  if (k == 20) return cc::light_level((int)w[32]);
  return (float)((w0 >> 24) & 1u);                                                // sleeping
}
PQN_D float cc_obs_elem(const int *cells, const float *tail, int i) {
  if (i >= 1323) return tail[i - 1323];
  const int cell = i / 21, ch = i - cell * 21;
  const int code = cells[cell];
  return ch < 17 ? ((code & 255) == ch ? 1.0f : 0.0f) : (((code >> (8 + ch - 17)) & 1) ? 1.0f : 0.0f);
}

// workgroup per env (any n, any alignment of obs)
__global__ __launch_bounds__(256) void cc_obs_kernel(int n, const uint32_t *__restrict__ state, float *__restrict__ obs) {
  __shared__ int s_cell[64];
  __shared__ float s_tail[22];
  // the scalar words of env e sit in SCALAR_WORDS different cache lines ([word][env] layout): ONE vector load (lane =
  // word), then every lane unpacks from LDS.  A uniform-address load per word compiles to a chain of scalar loads, each
  // waited for before the next.
  __shared__ uint32_t s_w[cc::SCALAR_WORDS];
  const int e = blockIdx.x, tid = threadIdx.x;
  if (tid < cc::SCALAR_WORDS) s_w[tid] = state[(size_t)(cc::MAP_WORDS + tid) * n + e];
  __syncthreads();
  if (tid < 63) s_cell[tid] = cc_obs_cell(s_w, state, n, e, tid);
  if (tid < 22) s_tail[tid] = cc_obs_tail(s_w, tid);
  __syncthreads();
  float *dst = obs + (size_t)e * cc::OBS;
  for (int i = tid; i < cc::OBS; i += 256) dst[i] = cc_obs_elem(s_cell, s_tail, i);
}

// four consecutive envs per workgroup (n % 4 == 0, obs 16-B aligned): their rows are ONE 16-B aligned block of 4 x 1345
// floats, written with 16-B stores.  Measured on the env-per-workgroup form at 1024 envs (rocprofv3, ablations): 22 us per
// launch, 17 of them in the 4-B stores to rows that start at any 4-B offset -- 5.5 MB at 0.3 TB/s.
__global__ __launch_bounds__(256) void cc_obs4_kernel(int n, const uint32_t *__restrict__ state, float *__restrict__ obs) {
  __shared__ uint32_t s_w[4][cc::SCALAR_WORDS + 1];
  __shared__ int s_cell[4][64];
  __shared__ float s_tail[4][24];
  const int tid = threadIdx.x, e0 = blockIdx.x * 4;
  if (tid < 4 * cc::SCALAR_WORDS) {   // lane = (word, env): the four envs of a word are 16 contiguous bytes
    const int wd = tid >> 2, le = tid & 3;
    s_w[le][wd] = state[(size_t)(cc::MAP_WORDS + wd) * n + e0 + le];
  }
  __syncthreads();
  if (tid < 4 * 63) {
    const int le = tid / 63, k = tid - 63 * le;
    s_cell[le][k] = cc_obs_cell(s_w[le], state, n, e0 + le, k);
  }
  if (tid < 4 * 22) {
    const int le = tid / 22, k = tid - 22 * le;
    s_tail[le][k] = cc_obs_tail(s_w[le], k);
  }
  __syncthreads();
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 *dst = reinterpret_cast<f4 *>(obs + (size_t)e0 * cc::OBS);
  for (int q = tid; q < cc::OBS; q += 256) {   // 4 envs x OBS floats = OBS 16-B packs
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = 4 * q + j, le = g / cc::OBS, i = g - le * cc::OBS;
      v[j] = cc_obs_elem(s_cell[le], s_tail[le], i);
    }
    dst[q] = f4{v[0], v[1], v[2], v[3]};
  }
}
void cc_launch_obs(int n, const uint32_t *state, float *obs, hipStream_t st) {
  if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(obs) & 15) == 0)
    hipLaunchKernelGGL(cc_obs4_kernel, dim3(n / 4), dim3(256), 0, st, n, state, obs);
  else
    hipLaunchKernelGGL(cc_obs_kernel, dim3(n), dim3(256), 0, st, n, state, obs);
}

// canonical export / import (tests, checkpoints)
__global__ void cc_canon_kernel(int n, int do_export, uint32_t *state, int32_t *si, float *sf, uint32_t *log) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int32_t *c = si + (size_t)e * cc::CANON_SI;
  int32_t *t = c + 4096;
  cc::Scalars s;
  if (do_export) {
    for (int cell = 0; cell < 4096; ++cell) c[cell] = (state[(size_t)(cell >> 2) * n + e] >> (8 * (cell & 3))) & 255;
    s.load(state, n, e);
    t[0] = s.pr; t[1] = s.pc; t[2] = s.dir; t[3] = s.health; t[4] = s.food; t[5] = s.drink; t[6] = s.energy; t[7] = s.sleeping;
    for (int i = 0; i < 12; ++i) t[8 + i] = s.inv[i];
    for (int i = 0; i < cc::NZ; ++i) { int32_t *q = t + 20 + 5 * i; q[0] = s.z[i].r; q[1] = s.z[i].c; q[2] = s.z[i].health; q[3] = s.z[i].cd; q[4] = s.z[i].mask; }
    for (int i = 0; i < cc::NC; ++i) { int32_t *q = t + 35 + 4 * i; q[0] = s.cow[i].r; q[1] = s.cow[i].c; q[2] = s.cow[i].health; q[3] = s.cow[i].mask; }
    for (int i = 0; i < cc::NS; ++i) { int32_t *q = t + 47 + 5 * i; q[0] = s.sk[i].r; q[1] = s.sk[i].c; q[2] = s.sk[i].health; q[3] = s.sk[i].cd; q[4] = s.sk[i].mask; }
    for (int i = 0; i < cc::NA; ++i) { int32_t *q = t + 57 + 4 * i; q[0] = s.ar[i].r; q[1] = s.ar[i].c; q[2] = s.ar[i].cd; q[3] = s.ar[i].mask; }
    for (int i = 0; i < cc::NP; ++i) { int32_t *q = t + 69 + 4 * i; q[0] = s.pl[i].r; q[1] = s.pl[i].c; q[2] = s.pl[i].age; q[3] = s.pl[i].mask; }
    t[109] = s.ach; t[110] = s.timestep;
    float *f = sf + (size_t)e * cc::CANON_SF;
    f[0] = s.recover; f[1] = s.hunger; f[2] = s.thirst; f[3] = s.fatigue;
    if (log) for (int i = 0; i < PQN_LOG_WORDS; ++i) log[(size_t)e * PQN_LOG_WORDS + i] = state[(size_t)(cc::ENV_WORDS + i) * n + e];
  } else {
    for (int mw = 0; mw < cc::MAP_WORDS; ++mw)
      state[(size_t)mw * n + e] = (uint32_t)c[4 * mw] | ((uint32_t)c[4 * mw + 1] << 8) | ((uint32_t)c[4 * mw + 2] << 16) | ((uint32_t)c[4 * mw + 3] << 24);
    s.pr = t[0]; s.pc = t[1]; s.dir = t[2]; s.health = t[3]; s.food = t[4]; s.drink = t[5]; s.energy = t[6]; s.sleeping = t[7];
    for (int i = 0; i < 12; ++i) s.inv[i] = t[8 + i];
    for (int i = 0; i < cc::NZ; ++i) { const int32_t *q = t + 20 + 5 * i; s.z[i] = cc::Mob{q[0], q[1], q[2], q[3], q[4]}; }
    for (int i = 0; i < cc::NC; ++i) { const int32_t *q = t + 35 + 4 * i; s.cow[i] = cc::Mob{q[0], q[1], q[2], 0, q[3]}; }
    for (int i = 0; i < cc::NS; ++i) { const int32_t *q = t + 47 + 5 * i; s.sk[i] = cc::Mob{q[0], q[1], q[2], q[3], q[4]}; }
    for (int i = 0; i < cc::NA; ++i) { const int32_t *q = t + 57 + 4 * i; s.ar[i] = cc::Mob{q[0], q[1], 0, q[2], q[3]}; }
    for (int i = 0; i < cc::NP; ++i) { const int32_t *q = t + 69 + 4 * i; s.pl[i] = cc::Plant{q[0], q[1], q[2], q[3]}; }
    s.ach = t[109]; s.timestep = t[110];
    const float *f = sf + (size_t)e * cc::CANON_SF;
    s.recover = f[0]; s.hunger = f[1]; s.thirst = f[2]; s.fatigue = f[3];
    s.store(state, n, e);
    for (int i = 0; i < PQN_LOG_WORDS; ++i) state[(size_t)(cc::ENV_WORDS + i) * n + e] = log ? log[(size_t)e * PQN_LOG_WORDS + i] : 0u;
  }
}

// ======================================================================================================================
// host side (called from pqn_env.hip's dispatch)
// ======================================================================================================================
void pqn_craftax_spec(pqn_env_spec_t *s) {
  s->obs_dim[0] = cc::OBS; s->obs_dim[1] = 0; s->obs_dim[2] = 0;
  s->obs_size = cc::OBS;
  s->num_actions = cc::NUM_ACTIONS;
  s->max_steps = cc::MAX_STEPS;
  s->state_words = cc::ENV_WORDS + PQN_LOG_WORDS;
  s->obs_words = 0;
  s->canon_si = cc::CANON_SI;
  s->canon_sf = cc::CANON_SF;
}

// scratch for the reset slots: one i32 per env behind the state words is not available (caller-owned layout), so the
// slots live in small device buffers owned by the library -- ONE PER STREAM (grown on demand; not on the per-step hot
// path): seeds that run as concurrent HIP streams (pqn.py: _vmap_streams) each call reset / step on their own stream, and a
// shared buffer would let one seed's cc_reset_kernel overwrite the slots another seed's cc_world_kernel is about to read.
#define CC_MAX_STREAMS 64
static struct { hipStream_t st; int32_t *p; int n; } g_cc_slots[CC_MAX_STREAMS];
static int g_cc_nstreams = 0;
static int32_t *cc_slots(int n, hipStream_t st) {
  int k = 0;
  while (k < g_cc_nstreams && g_cc_slots[k].st != st) ++k;
  if (k == g_cc_nstreams) {
    if (g_cc_nstreams == CC_MAX_STREAMS) {   // table full: recycle the first entry once its stream has drained
      k = 0;
      (void)hipStreamSynchronize(g_cc_slots[0].st);
    } else {
      ++g_cc_nstreams;
      g_cc_slots[k].p = nullptr;
      g_cc_slots[k].n = 0;
    }
    g_cc_slots[k].st = st;
  }
  if (n > g_cc_slots[k].n) {
    if (g_cc_slots[k].p) { (void)hipStreamSynchronize(st); (void)hipFree(g_cc_slots[k].p); }
    if (hipMalloc(&g_cc_slots[k].p, sizeof(int32_t) * (size_t)n) != hipSuccess) { g_cc_slots[k].p = nullptr; g_cc_slots[k].n = 0; return nullptr; }
    g_cc_slots[k].n = n;
  }
  return g_cc_slots[k].p;
}

int pqn_craftax_reset(int n, uint64_t key, uint32_t *state, float *obs, hipStream_t st) {
  int32_t *slots = cc_slots(n, st);
  PQN_REQUIRE(slots, "Craftax-Classic: cannot allocate the reset-slot scratch");
  const dim3 g((n + 255) / 256), b(256);
  hipLaunchKernelGGL(cc_reset_kernel, g, b, 0, st, n, 0, 1, (const uint8_t *)nullptr, (const uint64_t *)nullptr, state, slots);
  hipLaunchKernelGGL(cc_world_kernel, dim3(n, cc::MAP * cc::MAP / 256), b, 0, st, n, key, (const uint64_t *)nullptr, 0, slots, state);
  if (obs) cc_launch_obs(n, state, obs, st);
  return pqn_check_launch("pqn_env_reset(Craftax-Classic)");
}

// reset_ratio == 0: gymnax-style auto-reset; > 0: OptimisticResetVecEnvWrapper semantics (scratch = u64[n])
int pqn_craftax_step(int n, uint64_t key, const uint64_t *key_dev, float rscale, uint32_t *state, const int32_t *action,
                     const pqn_step_out_t &out, int reset_ratio, uint64_t *scratch, int32_t *slot_out, hipStream_t st) {
  int32_t *slots = slot_out ? slot_out : cc_slots(n, st);
  PQN_REQUIRE(slots, "Craftax-Classic: cannot allocate the reset-slot scratch");
  PQN_REQUIRE(out.obs_bits == nullptr, "Craftax-Classic has no packed observation");
  const dim3 g((n + 255) / 256), b(256);
  hipLaunchKernelGGL(cc_step_kernel, g, b, 0, st, n, key, key_dev, rscale, state, action, out, reset_ratio > 0 ? scratch : (uint64_t *)nullptr);
  hipLaunchKernelGGL(cc_reset_kernel, g, b, 0, st, n, reset_ratio > 0 ? 2 : 1, reset_ratio > 0 ? reset_ratio : 1, out.done, scratch, state, slots);
  hipLaunchKernelGGL(cc_world_kernel, dim3(n, cc::MAP * cc::MAP / 256), b, 0, st, n, key, key_dev, reset_ratio > 0 ? 1 : 0, slots, state);
  if (out.obs) cc_launch_obs(n, state, out.obs, st);
  return pqn_check_launch("pqn_env_step(Craftax-Classic)");
}

int pqn_craftax_canon(int n, int do_export, uint32_t *state, int32_t *si, float *sf, uint32_t *log, hipStream_t st) {
  PQN_REQUIRE(sf, "Craftax-Classic canonical state needs sf");
  hipLaunchKernelGGL(cc_canon_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, do_export, state, si, sf, log);
  return pqn_check_launch("pqn_env_canon(Craftax-Classic)");
}
