// pqn_env_rules.h -- per-environment transition rules as device structs (packed SoA state), shared by
// the env kernels (pqn_env.hip) and the persistent rollout kernel (pqn_qnet.hip).
// Rules follow gymnax==0.0.6 / MinAtar (Young & Tian 2019); see oracle/pqn_oracle.h on parity status.
#pragma once
#include "pqn_common.h"

// ===========================================================================
// Breakout-MinAtar.  2 state words:
//  w0 = bricks[30] (bit (y-1)*10+x, rows 1..3) | strike<<30 | terminal<<31
//  w1 = ball_x | ball_y<<4 | dir<<8 | pos<<10 | last_x<<14 | last_y<<18 | time<<22
// ===========================================================================
struct Breakout {
  static constexpr int ENV_WORDS = 2;
  static constexpr int OBS_C = 4;
  static constexpr int OBS_SIZE = 400;
  static constexpr int OBS_WORDS = 16;  // 13 used, padded to 16 B multiple
  static constexpr int NUM_ACTIONS = 3;
  static constexpr int MAX_STEPS = 1000;
  static constexpr int CANON_SI = 109;
  static constexpr int CANON_SF = 0;
  static constexpr uint32_t FULL = 0x3FFFFFFFu;

  uint32_t bricks;
  int ball_x, ball_y, dir, pos, last_x, last_y, time;
  int strike, terminal;

  PQN_D void unpack(const uint32_t *w) {
    bricks = w[0] & FULL;
    strike = (w[0] >> 30) & 1;
    terminal = (w[0] >> 31) & 1;
    ball_x = w[1] & 15;
    ball_y = (w[1] >> 4) & 15;
    dir = (w[1] >> 8) & 3;
    pos = (w[1] >> 10) & 15;
    last_x = (w[1] >> 14) & 15;
    last_y = (w[1] >> 18) & 15;
    time = (w[1] >> 22) & 1023;
  }
  PQN_D void pack(uint32_t *w) const {
    w[0] = bricks | ((uint32_t)strike << 30) | ((uint32_t)terminal << 31);
    w[1] = (uint32_t)ball_x | ((uint32_t)ball_y << 4) | ((uint32_t)dir << 8) | ((uint32_t)pos << 10) |
           ((uint32_t)last_x << 14) | ((uint32_t)last_y << 18) | ((uint32_t)time << 22);
  }
  PQN_D void reset(uint64_t key, uint32_t e) {
    uint32_t o0, o1;
    pqn_bits(key, e, PQN_STREAM_RESET, o0, o1);
    const int start = (int)(o0 & 1u);
    bricks = FULL;
    strike = 0;
    terminal = 0;
    ball_y = 3;
    ball_x = start ? 9 : 0;
    dir = start ? 3 : 2;
    pos = 4;
    last_y = 3;
    last_x = ball_x;
    time = 0;
  }
  PQN_D bool brick_at(int y, int x) const {
    const unsigned r = (unsigned)(y - 1);
    return r < 3u && ((bricks >> (r * 10 + x)) & 1u);
  }
  // step_env; returns reward, sets done.  key unused (deterministic rules).
  PQN_D float step(int action, uint64_t, uint32_t, int &done) {
    float reward = 0.0f;
    if (action == 1) pos = max(0, pos - 1);
    else if (action == 2) pos = min(9, pos + 1);
    const int ox = ball_x, oy = ball_y;
    int nx = ox + ((dir == 1 || dir == 2) ? 1 : -1);
    int ny = oy + ((dir >= 2) ? 1 : -1);
    if (nx < 0 || nx > 9) {
      nx = nx < 0 ? 0 : 9;
      dir ^= 1;  // [1,0,3,2]
    }
    int term = 0, toggle = 0;
    if (ny < 0) {
      ny = 0;
      dir ^= 3;  // [3,2,1,0]
    } else if (brick_at(ny, nx)) {
      toggle = 1;
      if (!strike) {
        reward = 1.0f;
        bricks &= ~(1u << ((ny - 1) * 10 + nx));
        ny = oy;
        dir ^= 3;
      }
    } else if (ny == 9) {
      if (bricks == 0u) bricks = FULL;
      if (ox == pos) {
        dir ^= 3;
        ny = oy;
      } else if (nx == pos) {
        dir ^= 2;  // [2,3,0,1]
        ny = oy;
      } else {
        term = 1;
      }
    }
    strike = toggle;
    last_x = ox;
    last_y = oy;
    ball_x = nx;
    ball_y = ny;
    time += 1;
    done = term | (time >= MAX_STEPS);
    terminal = done;
    return reward;
  }
  // bit (cell*4 + c): c0 paddle, c1 ball, c2 trail, c3 brick
  PQN_D void obs_bits(uint32_t *o) const {
#pragma unroll
    for (int i = 0; i < OBS_WORDS; ++i) o[i] = 0u;
    // bricks occupy cells 10..39 -> nibble bit 3 of words 1..4
#pragma unroll
    for (int w = 1; w <= 4; ++w) {
      const int c0 = w * 8 - 10;  // first brick index covered by this word (may be negative)
      uint32_t b = c0 >= 0 ? (bricks >> c0) : (bricks << (-c0));
      b &= 0xFFu;
      b = (b | (b << 12)) & 0x000F000Fu;
      b = (b | (b << 6)) & 0x03030303u;
      b = (b | (b << 3)) & 0x11111111u;
      o[w] = b << 3;
    }
    set(o, (90 + pos) * 4 + 0);
    set(o, (ball_y * 10 + ball_x) * 4 + 1);
    set(o, (last_y * 10 + last_x) * 4 + 2);
  }
  static PQN_D void set(uint32_t *o, int bit) { o[bit >> 5] |= 1u << (bit & 31); }
  PQN_D void to_canon(int32_t *si, float *) const {
    si[0] = ball_y; si[1] = ball_x; si[2] = dir; si[3] = pos; si[4] = strike;
    si[5] = last_y; si[6] = last_x; si[7] = time; si[8] = terminal;
    for (int c = 0; c < 100; ++c) si[9 + c] = brick_at(c / 10, c % 10) ? 1 : 0;
  }
  PQN_D void from_canon(const int32_t *si, const float *) {
    ball_y = si[0]; ball_x = si[1]; dir = si[2]; pos = si[3]; strike = si[4];
    last_y = si[5]; last_x = si[6]; time = si[7]; terminal = si[8];
    bricks = 0u;
    for (int c = 10; c < 40; ++c)
      if (si[9 + c]) bricks |= 1u << (c - 10);
  }
};


// helper: set observation bit (cell, channel) in a packed row
template <int C>
PQN_D void obs_set(uint32_t *o, int cell, int c) {
  const int bit = cell * C + c;
  o[bit >> 5] |= 1u << (bit & 31);
}

// ===========================================================================
// Asterix-MinAtar (MinAtar asterix.py rules).  4 state words:
//  w0/w1 = entities 0..3 / 4..7, 8 bits each: x | present<<4 | moves_right<<5 | is_gold<<6  (row = slot+1)
//  w2 = player_x | player_y<<4 | spawn_speed<<8 | spawn_timer<<12 | move_speed<<16 | move_timer<<19
//       | ramp_index<<22 | terminal<<28
//  w3 = (ramp_timer+1) | time<<7
// ===========================================================================
struct Asterix {
  static constexpr int ENV_WORDS = 4;
  static constexpr int OBS_SIZE = 400;
  static constexpr int OBS_WORDS = 16;
  static constexpr int NUM_ACTIONS = 5;
  static constexpr int MAX_STEPS = 1000;
  static constexpr int CANON_SI = 43;
  static constexpr int CANON_SF = 0;
  uint32_t ent[8];  // x | present<<4 | lr<<5 | gold<<6
  int px, py, spawn_speed, spawn_timer, move_speed, move_timer, ramp_timer, ramp_index, time, terminal;

  PQN_D void unpack(const uint32_t *w) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ent[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
    px = w[2] & 15; py = (w[2] >> 4) & 15; spawn_speed = (w[2] >> 8) & 15; spawn_timer = (w[2] >> 12) & 15;
    move_speed = (w[2] >> 16) & 7; move_timer = (w[2] >> 19) & 7; ramp_index = (w[2] >> 22) & 63;
    terminal = (w[2] >> 28) & 1;
    ramp_timer = (int)(w[3] & 127) - 1;
    time = (w[3] >> 7) & 1023;
  }
  PQN_D void pack(uint32_t *w) const {
    w[0] = ent[0] | (ent[1] << 8) | (ent[2] << 16) | (ent[3] << 24);
    w[1] = ent[4] | (ent[5] << 8) | (ent[6] << 16) | (ent[7] << 24);
    w[2] = (uint32_t)px | ((uint32_t)py << 4) | ((uint32_t)spawn_speed << 8) | ((uint32_t)spawn_timer << 12) |
           ((uint32_t)move_speed << 16) | ((uint32_t)move_timer << 19) | ((uint32_t)ramp_index << 22) |
           ((uint32_t)terminal << 28);
    w[3] = (uint32_t)(ramp_timer + 1) | ((uint32_t)time << 7);
  }
  PQN_D void reset(uint64_t, uint32_t) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ent[i] = 0u;
    px = 5; py = 5; spawn_speed = 10; spawn_timer = 10; move_speed = 5; move_timer = 5; ramp_timer = 100;
    ramp_index = 0; time = 0; terminal = 0;
  }
  PQN_D float step(int action, uint64_t key, uint32_t e, int &done) {
    float r = 0.0f;
    int term = 0;
    if (spawn_timer == 0) {
      uint32_t o0, o1, p0, p1;
      pqn_bits(key, e, PQN_STREAM_ENV, o0, o1);
      pqn_bits(key, e, PQN_STREAM_ENV + 1, p0, p1);
      const uint32_t lr = o0 & 1u;
      const uint32_t gold = pqn_uniform(o1) < (1.0f / 3.0f) ? 1u : 0u;
      int nfree = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) nfree += !(ent[i] & 16u);
      if (nfree > 0) {
        int k = (int)pqn_randint(p0, (uint32_t)nfree);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool free_slot = !(ent[i] & 16u);
          if (free_slot && k == 0) ent[i] = (lr ? 0u : 9u) | 16u | (lr << 5) | (gold << 6);
          k -= free_slot ? 1 : 0;
        }
      }
      spawn_timer = spawn_speed;
    }
    if (action == 1) px = max(0, px - 1);
    else if (action == 3) px = min(9, px + 1);
    else if (action == 2) py = max(1, py - 1);
    else if (action == 4) py = min(8, py + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((ent[i] & 16u) && (int)(ent[i] & 15u) == px && i + 1 == py) {
        if (ent[i] & 64u) { ent[i] = 0u; r += 1.0f; } else term = 1;
      }
    }
    if (move_timer == 0) {
      move_timer = move_speed;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (!(ent[i] & 16u)) continue;
        const int x = (int)(ent[i] & 15u) + ((ent[i] & 32u) ? 1 : -1);
        if (x < 0 || x > 9) { ent[i] = 0u; continue; }
        ent[i] = (ent[i] & ~15u) | (uint32_t)x;
        if (x == px && i + 1 == py) {
          if (ent[i] & 64u) { ent[i] = 0u; r += 1.0f; } else term = 1;
        }
      }
    }
    spawn_timer -= 1;
    move_timer -= 1;
    if (spawn_speed > 1 || move_speed > 1) {
      if (ramp_timer >= 0) ramp_timer -= 1;
      else {
        if (move_speed > 1 && (ramp_index & 1)) move_speed -= 1;
        if (spawn_speed > 1) spawn_speed -= 1;
        ramp_index += 1;
        ramp_timer = 100;
      }
    }
    time += 1;
    done = term | (time >= MAX_STEPS);
    terminal = done;
    return r;
  }
  PQN_D void obs_bits(uint32_t *o) const {
#pragma unroll
    for (int i = 0; i < OBS_WORDS; ++i) o[i] = 0u;
    obs_set<4>(o, py * 10 + px, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (!(ent[i] & 16u)) continue;
      const int x = (int)(ent[i] & 15u);
      obs_set<4>(o, (i + 1) * 10 + x, (ent[i] & 64u) ? 3 : 1);
      const int back = (ent[i] & 32u) ? x - 1 : x + 1;
      if (back >= 0 && back <= 9) obs_set<4>(o, (i + 1) * 10 + back, 2);
    }
  }
  PQN_D void to_canon(int32_t *si, float *) const {
    si[0] = px; si[1] = py; si[2] = 0; si[3] = spawn_speed; si[4] = spawn_timer; si[5] = move_speed;
    si[6] = move_timer; si[7] = ramp_timer; si[8] = ramp_index; si[9] = time; si[10] = terminal;
    for (int i = 0; i < 8; ++i) {
      const bool p = ent[i] & 16u;
      si[11 + 4 * i] = p ? (int)(ent[i] & 15u) : 0;
      si[11 + 4 * i + 1] = p ? 1 : 0;
      si[11 + 4 * i + 2] = p ? (int)((ent[i] >> 5) & 1u) : 0;
      si[11 + 4 * i + 3] = p ? (int)((ent[i] >> 6) & 1u) : 0;
    }
  }
  PQN_D void from_canon(const int32_t *si, const float *) {
    px = si[0]; py = si[1]; spawn_speed = si[3]; spawn_timer = si[4]; move_speed = si[5]; move_timer = si[6];
    ramp_timer = si[7]; ramp_index = si[8]; time = si[9]; terminal = si[10];
    for (int i = 0; i < 8; ++i)
      ent[i] = si[11 + 4 * i + 1] ? ((uint32_t)si[11 + 4 * i] | 16u | ((uint32_t)si[11 + 4 * i + 2] << 5) |
                                     ((uint32_t)si[11 + 4 * i + 3] << 6)) : 0u;
  }
};

// ===========================================================================
// Freeway-MinAtar (MinAtar freeway.py rules).  5 state words:
//  w0..w3 = 2 cars each, 11 bits per car: x | timer<<4 | (speed+5)<<7   (row = car+1)
//  w4 = pos | move_timer<<4 | terminate_timer<<6 | time<<18 | terminal<<30
// ===========================================================================
struct Freeway {
  static constexpr int ENV_WORDS = 5;
  static constexpr int OBS_SIZE = 700;
  static constexpr int OBS_WORDS = 24;
  static constexpr int NUM_ACTIONS = 3;
  static constexpr int MAX_STEPS = 2500;
  static constexpr int CANON_SI = 29;
  static constexpr int CANON_SF = 0;
  int cx[8], ct[8], cs[8];
  int pos, move_timer, terminate_timer, time, terminal;

  PQN_D void unpack(const uint32_t *w) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t c = (w[i >> 1] >> (11 * (i & 1))) & 0x7FFu;
      cx[i] = c & 15; ct[i] = (c >> 4) & 7; cs[i] = (int)((c >> 7) & 15) - 5;
    }
    pos = w[4] & 15; move_timer = (w[4] >> 4) & 3; terminate_timer = (w[4] >> 6) & 4095; time = (w[4] >> 18) & 4095;
    terminal = (w[4] >> 30) & 1;
  }
  PQN_D void pack(uint32_t *w) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t a = (uint32_t)cx[2 * j] | ((uint32_t)ct[2 * j] << 4) | ((uint32_t)(cs[2 * j] + 5) << 7);
      const uint32_t b = (uint32_t)cx[2 * j + 1] | ((uint32_t)ct[2 * j + 1] << 4) | ((uint32_t)(cs[2 * j + 1] + 5) << 7);
      w[j] = a | (b << 11);
    }
    w[4] = (uint32_t)pos | ((uint32_t)move_timer << 4) | ((uint32_t)terminate_timer << 6) | ((uint32_t)time << 18) |
           ((uint32_t)terminal << 30);
  }
  PQN_D void randomize(uint64_t key, uint32_t e, uint32_t stream0, bool initialize) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t o0, o1;
      pqn_bits(key, e, stream0 + (uint32_t)i, o0, o1);
      const int speed = 1 + (int)pqn_randint(o0, 5u);
      if (initialize) cx[i] = 0;
      ct[i] = speed;
      cs[i] = (o1 & 1u) ? speed : -speed;
    }
  }
  PQN_D void reset(uint64_t key, uint32_t e) {
    randomize(key, e, 16u, true);
    pos = 9; move_timer = 3; terminate_timer = 2500; time = 0; terminal = 0;
  }
  PQN_D float step(int action, uint64_t key, uint32_t e, int &done) {
    float r = 0.0f;
    if (action == 1 && move_timer == 0) { move_timer = 3; pos = max(0, pos - 1); }
    else if (action == 2 && move_timer == 0) { move_timer = 3; pos = min(9, pos + 1); }
    if (pos == 0) { r += 1.0f; randomize(key, e, 32u, false); pos = 9; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (cx[i] == 4 && i + 1 == pos) pos = 9;
      if (ct[i] == 0) {
        ct[i] = abs(cs[i]);
        cx[i] += cs[i] > 0 ? 1 : -1;
        if (cx[i] < 0) cx[i] = 9; else if (cx[i] > 9) cx[i] = 0;
        if (cx[i] == 4 && i + 1 == pos) pos = 9;
      } else ct[i] -= 1;
    }
    move_timer -= move_timer > 0 ? 1 : 0;
    terminate_timer -= 1;
    const int term = terminate_timer < 0;
    time += 1;
    done = term | (time >= MAX_STEPS);
    terminal = done;
    if (terminate_timer < 0) terminate_timer = 0;  // only reachable together with done (reset follows)
    return r;
  }
  PQN_D void obs_bits(uint32_t *o) const {
#pragma unroll
    for (int i = 0; i < OBS_WORDS; ++i) o[i] = 0u;
    obs_set<7>(o, pos * 10 + 4, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      obs_set<7>(o, (i + 1) * 10 + cx[i], 1);
      int back = cs[i] > 0 ? cx[i] - 1 : cx[i] + 1;
      if (back < 0) back = 9; else if (back > 9) back = 0;
      obs_set<7>(o, (i + 1) * 10 + back, 1 + abs(cs[i]));
    }
  }
  PQN_D void to_canon(int32_t *si, float *) const {
    si[0] = pos; si[1] = move_timer; si[2] = terminate_timer; si[3] = time; si[4] = terminal;
    for (int i = 0; i < 8; ++i) { si[5 + 3 * i] = cx[i]; si[5 + 3 * i + 1] = ct[i]; si[5 + 3 * i + 2] = cs[i]; }
  }
  PQN_D void from_canon(const int32_t *si, const float *) {
    pos = si[0]; move_timer = si[1]; terminate_timer = si[2]; time = si[3]; terminal = si[4];
    for (int i = 0; i < 8; ++i) { cx[i] = si[5 + 3 * i]; ct[i] = si[5 + 3 * i + 1]; cs[i] = si[5 + 3 * i + 2]; }
  }
};

// ===========================================================================
// SpaceInvaders-MinAtar (MinAtar space_invaders.py rules).  14 state words:
//  w0..3 alien map, w4..7 friendly bullets, w8..11 enemy bullets: row r = 10 bits at (r%3)*10 of word r/3
//  w12 = pos | (dir>0)<<4 | enemy_move_interval<<5 | alien_move_timer<<9 | alien_shot_timer<<13
//        | shot_timer<<17 | ramp_index<<20 | terminal<<24        w13 = time
// ===========================================================================
struct SpaceInvaders {
  static constexpr int ENV_WORDS = 14;
  static constexpr int OBS_SIZE = 600;
  static constexpr int OBS_WORDS = 20;
  static constexpr int NUM_ACTIONS = 4;
  static constexpr int MAX_STEPS = 1000;
  static constexpr int CANON_SI = 309;
  static constexpr int CANON_SF = 0;
  uint32_t al[10], fb[10], eb[10];  // row bitmasks, bit x = column x
  int pos, dir, interval, move_timer, shot_timer_alien, shot_timer, ramp_index, time, terminal;

  static PQN_D void unpack_map(const uint32_t *w, uint32_t *rows) {
#pragma unroll
    for (int r = 0; r < 10; ++r) rows[r] = (w[r / 3] >> (10 * (r % 3))) & 0x3FFu;
  }
  static PQN_D void pack_map(const uint32_t *rows, uint32_t *w) {
    w[0] = rows[0] | (rows[1] << 10) | (rows[2] << 20);
    w[1] = rows[3] | (rows[4] << 10) | (rows[5] << 20);
    w[2] = rows[6] | (rows[7] << 10) | (rows[8] << 20);
    w[3] = rows[9];
  }
  PQN_D void unpack(const uint32_t *w) {
    unpack_map(w, al); unpack_map(w + 4, fb); unpack_map(w + 8, eb);
    pos = w[12] & 15; dir = ((w[12] >> 4) & 1) ? 1 : -1; interval = (w[12] >> 5) & 15; move_timer = (w[12] >> 9) & 15;
    shot_timer_alien = (w[12] >> 13) & 15; shot_timer = (w[12] >> 17) & 7; ramp_index = (w[12] >> 20) & 15;
    terminal = (w[12] >> 24) & 1;
    time = (int)w[13];
  }
  PQN_D void pack(uint32_t *w) const {
    pack_map(al, w); pack_map(fb, w + 4); pack_map(eb, w + 8);
    w[12] = (uint32_t)pos | ((dir > 0 ? 1u : 0u) << 4) | ((uint32_t)interval << 5) | ((uint32_t)move_timer << 9) |
            ((uint32_t)shot_timer_alien << 13) | ((uint32_t)shot_timer << 17) | ((uint32_t)ramp_index << 20) |
            ((uint32_t)terminal << 24);
    w[13] = (uint32_t)time;
  }
  PQN_D void fill_aliens() {
#pragma unroll
    for (int r = 0; r < 4; ++r) al[r] = 0xFCu;  // columns 2..7
  }
  PQN_D void reset(uint64_t, uint32_t) {
#pragma unroll
    for (int r = 0; r < 10; ++r) { al[r] = 0u; fb[r] = 0u; eb[r] = 0u; }
    fill_aliens();
    pos = 5; dir = -1; interval = 12; move_timer = 12; shot_timer_alien = 10; shot_timer = 0; ramp_index = 0;
    time = 0; terminal = 0;
  }
  PQN_D int count_aliens() const {
    int n = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) n += __popc(al[r]);
    return n;
  }
  PQN_D float step(int action, uint64_t, uint32_t, int &done) {
    float rew = 0.0f;
    int term = 0;
    if (action == 3 && shot_timer == 0) { fb[9] |= 1u << pos; shot_timer = 5; }
    else if (action == 1) pos = max(0, pos - 1);
    else if (action == 2) pos = min(9, pos + 1);
#pragma unroll
    for (int r = 0; r < 9; ++r) fb[r] = fb[r + 1];
    fb[9] = 0u;
#pragma unroll
    for (int r = 9; r > 0; --r) eb[r] = eb[r - 1];
    eb[0] = 0u;
    if ((eb[9] >> pos) & 1u) term = 1;
    if ((al[9] >> pos) & 1u) term = 1;
    if (move_timer == 0) {
      move_timer = min(count_aliens(), interval);
      uint32_t any = 0u;
#pragma unroll
      for (int r = 0; r < 10; ++r) any |= al[r];
      if (((any & 1u) && dir < 0) || ((any & 0x200u) && dir > 0)) {
        dir = -dir;
        if (al[9]) term = 1;
        const uint32_t last = al[9];
#pragma unroll
        for (int r = 9; r > 0; --r) al[r] = al[r - 1];
        al[0] = last;  // np.roll wraps
      } else {
#pragma unroll
        for (int r = 0; r < 10; ++r)
          al[r] = dir > 0 ? (((al[r] << 1) | (al[r] >> 9)) & 0x3FFu) : (((al[r] >> 1) | (al[r] << 9)) & 0x3FFu);
      }
      if ((al[9] >> pos) & 1u) term = 1;
    }
    if (shot_timer_alien == 0) {
      shot_timer_alien = 10;
      // nearest alien column by |x - pos| (ties: smaller x), lowest alien in it
      bool found = false;
      for (int d = 0; d < 10 && !found; ++d) {
        for (int sg = -1; sg <= 1 && !found; sg += 2) {
          if (d == 0 && sg == 1) continue;
          const int x = pos + sg * d;
          if (x < 0 || x > 9) continue;
          int ymax = -1;
#pragma unroll
          for (int r = 0; r < 10; ++r)
            if ((al[r] >> x) & 1u) ymax = r;
          if (ymax >= 0) {
#pragma unroll
            for (int r = 0; r < 10; ++r)
              if (r == ymax) eb[r] |= 1u << x;
            found = true;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint32_t kill = al[r] & fb[r];
      rew += (float)__popc(kill);
      al[r] &= ~kill;
      fb[r] &= ~kill;
    }
    shot_timer -= shot_timer > 0 ? 1 : 0;
    move_timer -= 1;
    shot_timer_alien -= 1;
    const int cnt = count_aliens();
    if (interval > 6 && cnt == 0) { interval -= 1; ramp_index += 1; }
    if (cnt == 0) fill_aliens();
    time += 1;
    done = term | (time >= MAX_STEPS);
    terminal = done;
    return rew;
  }
  PQN_D void obs_bits(uint32_t *o) const {
#pragma unroll
    for (int i = 0; i < OBS_WORDS; ++i) o[i] = 0u;
    obs_set<6>(o, 90 + pos, 0);
    for (int r = 0; r < 10; ++r) {
      for (int x = 0; x < 10; ++x) {
        const int cell = r * 10 + x;
        if ((al[r] >> x) & 1u) { obs_set<6>(o, cell, 1); obs_set<6>(o, cell, dir < 0 ? 2 : 3); }
        if ((fb[r] >> x) & 1u) obs_set<6>(o, cell, 4);
        if ((eb[r] >> x) & 1u) obs_set<6>(o, cell, 5);
      }
    }
  }
  PQN_D void to_canon(int32_t *si, float *) const {
    si[0] = pos; si[1] = dir; si[2] = interval; si[3] = move_timer; si[4] = shot_timer_alien; si[5] = shot_timer;
    si[6] = ramp_index; si[7] = time; si[8] = terminal;
    for (int c = 0; c < 100; ++c) {
      si[9 + c] = (al[c / 10] >> (c % 10)) & 1u;
      si[109 + c] = (fb[c / 10] >> (c % 10)) & 1u;
      si[209 + c] = (eb[c / 10] >> (c % 10)) & 1u;
    }
  }
  PQN_D void from_canon(const int32_t *si, const float *) {
    pos = si[0]; dir = si[1]; interval = si[2]; move_timer = si[3]; shot_timer_alien = si[4]; shot_timer = si[5];
    ramp_index = si[6]; time = si[7]; terminal = si[8];
    for (int r = 0; r < 10; ++r) { al[r] = 0u; fb[r] = 0u; eb[r] = 0u; }
    for (int c = 0; c < 100; ++c) {
      if (si[9 + c]) al[c / 10] |= 1u << (c % 10);
      if (si[109 + c]) fb[c / 10] |= 1u << (c % 10);
      if (si[209 + c]) eb[c / 10] |= 1u << (c % 10);
    }
  }
};

// ===========================================================================
// sin / cos for the classic-control dynamics, as explicit f32 arithmetic: Cody-Waite reduction by pi/2 in three
// pieces + the Cephes sinf / cosf minimax polynomials on [-pi/4, pi/4] (< 1 ulp-ish of libm on that range).  Device
// sinf / cosf and the host libm differ in the last ulp, which made GPU / CPU CartPole trajectories part ways after a
// few hundred steps; this sequence of IEEE mul / add / sub (no contraction: -ffp-contract=off) and one
// round-to-nearest-even gives the SAME bits on both sides (the oracle restates the identical sequence), so the
// CartPole parity tests are bit-exact trajectory tests.  gymnax's jnp.sin / jnp.cos differ from it by <= 1 ulp.
// ===========================================================================
PQN_HD void pqn_sincos_f32(float x, float &s, float &c) {
  const float k = __builtin_rintf(x * 0.636619772367581343f);
  float r = x - k * 1.5703125f;
  r = r - k * 4.837512969970703125e-4f;
  r = r - k * 7.54978995489188216e-8f;
  const float z = r * r;
  const float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  const float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  const int q = ((int)k) & 3;
  s = (q == 0) ? sp : (q == 1) ? cp : (q == 2) ? -sp : -cp;
  c = (q == 0) ? cp : (q == 1) ? -sp : (q == 2) ? -cp : sp;
}

// ===========================================================================
// CartPole-v1.  5 state words: x, x_dot, theta, theta_dot (f32 bits), time.
// ===========================================================================
struct CartPole {
  static constexpr int ENV_WORDS = 5;
  static constexpr int OBS_SIZE = 4;
  static constexpr int OBS_WORDS = 0;
  static constexpr int NUM_ACTIONS = 2;
  static constexpr int MAX_STEPS = 500;
  static constexpr int CANON_SI = 1;
  static constexpr int CANON_SF = 4;

  float x, x_dot, theta, theta_dot;
  int time;

  PQN_D void unpack(const uint32_t *w) {
    x = __uint_as_float(w[0]); x_dot = __uint_as_float(w[1]);
    theta = __uint_as_float(w[2]); theta_dot = __uint_as_float(w[3]);
    time = (int)w[4];
  }
  PQN_D void pack(uint32_t *w) const {
    w[0] = __float_as_uint(x); w[1] = __float_as_uint(x_dot);
    w[2] = __float_as_uint(theta); w[3] = __float_as_uint(theta_dot);
    w[4] = (uint32_t)time;
  }
  PQN_D void reset(uint64_t key, uint32_t e) {
    uint32_t a0, a1, b0, b1;
    pqn_bits(key, e, PQN_STREAM_RESET, a0, a1);
    pqn_bits(key, e, PQN_STREAM_RESET2, b0, b1);
    x = pqn_uniform(a0) * 0.1f - 0.05f;
    x_dot = pqn_uniform(a1) * 0.1f - 0.05f;
    theta = pqn_uniform(b0) * 0.1f - 0.05f;
    theta_dot = pqn_uniform(b1) * 0.1f - 0.05f;
    time = 0;
  }
  PQN_D int is_terminal() const {
    const float x_thr = 2.4f;
    const float th_thr = (float)(12.0 * 2.0 * 3.14159265358979323846 / 360.0);
    return (x < -x_thr) | (x > x_thr) | (theta < -th_thr) | (theta > th_thr) | (time >= MAX_STEPS);
  }
  PQN_D float step(int action, uint64_t, uint32_t, int &done) {
    const float gravity = 9.8f, masspole = 0.1f, total_mass = 1.1f, length = 0.5f;
    const float polemass_length = 0.05f, force_mag = 10.0f, tau = 0.02f;
    const int prev_terminal = is_terminal();
    const float force = force_mag * (float)action - force_mag * (float)(1 - action);
    float sintheta, costheta;
    pqn_sincos_f32(theta, sintheta, costheta);
    const float temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    const float thetaacc = (gravity * sintheta - costheta * temp) /
                           (length * (4.0f / 3.0f - masspole * (costheta * costheta) / total_mass));
    const float xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    const float nx = x + tau * x_dot;
    const float nxd = x_dot + tau * xacc;
    const float nt = theta + tau * theta_dot;
    const float ntd = theta_dot + tau * thetaacc;
    x = nx; x_dot = nxd; theta = nt; theta_dot = ntd;
    time += 1;
    done = is_terminal();
    return 1.0f - (float)prev_terminal;
  }
  PQN_D void obs_f32(float *o) const { o[0] = x; o[1] = x_dot; o[2] = theta; o[3] = theta_dot; }
  PQN_D void to_canon(int32_t *si, float *sf) const {
    si[0] = time; sf[0] = x; sf[1] = x_dot; sf[2] = theta; sf[3] = theta_dot;
  }
  PQN_D void from_canon(const int32_t *si, const float *sf) {
    time = si[0]; x = sf[0]; x_dot = sf[1]; theta = sf[2]; theta_dot = sf[3];
  }
};

// ===========================================================================
// Acrobot-v1 (gymnax 0.0.6 environments/classic_control/acrobot.py [3P-RECALL]; the alternative env named in the
// reference's config/alg/pqn_cartpole.yaml:24).  "Book" dynamics, one RK4 step of dt = 0.2, torque in {-1, 0, +1}, no
// torque noise, angles wrapped to [-pi, pi), velocities clipped to 4 pi / 9 pi, reward -1 until
// -cos(t1) - cos(t1 + t2) > 1, 500 steps.  Every expression is the oracle's (oracle/pqn_oracle.c acrobot_*), operation
// by operation, on the shared pqn_sincos_f32: bit-exact trajectories.
// ===========================================================================
struct Acrobot {
  static constexpr int ENV_WORDS = 5;
  static constexpr int OBS_SIZE = 6;
  static constexpr int OBS_WORDS = 0;
  static constexpr int NUM_ACTIONS = 3;
  static constexpr int MAX_STEPS = 500;
  static constexpr int CANON_SI = 1;
  static constexpr int CANON_SF = 4;

  float y[4];   // theta1, theta2, dtheta1, dtheta2
  int time;

  PQN_D void unpack(const uint32_t *w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = __uint_as_float(w[i]);
    time = (int)w[4];
  }
  PQN_D void pack(uint32_t *w) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = __float_as_uint(y[i]);
    w[4] = (uint32_t)time;
  }
  PQN_D void reset(uint64_t key, uint32_t e) {
    uint32_t a0, a1, b0, b1;
    pqn_bits(key, e, PQN_STREAM_RESET, a0, a1);
    pqn_bits(key, e, PQN_STREAM_RESET2, b0, b1);
    y[0] = pqn_uniform(a0) * 0.2f - 0.1f;
    y[1] = pqn_uniform(a1) * 0.2f - 0.1f;
    y[2] = pqn_uniform(b0) * 0.2f - 0.1f;
    y[3] = pqn_uniform(b1) * 0.2f - 0.1f;
    time = 0;
  }
  static PQN_D void dsdt(const float *q, float a, float *dq) {
    const float half_pi = 1.57079632679489661923f;
    const float t1 = q[0], t2 = q[1], w1 = q[2], w2 = q[3];
    float s2, c2, su, c12, c1;
    pqn_sincos_f32(t2, s2, c2);
    pqn_sincos_f32((t1 + t2) - half_pi, su, c12);
    pqn_sincos_f32(t1 - half_pi, su, c1);
    const float d1 = (0.25f + (1.25f + c2)) + 2.0f;
    const float d2 = (0.25f + 0.5f * c2) + 1.0f;
    const float phi2 = 4.9f * c12;
    const float phi1 = (((-0.5f * (w2 * w2)) * s2 - ((1.0f * w2) * w1) * s2) + 14.7f * c1) + phi2;
    const float dd2 = (((a + (d2 / d1) * phi1) - (0.5f * (w1 * w1)) * s2) - phi2) / (1.25f - (d2 * d2) / d1);
    const float dd1 = -((d2 * dd2 + phi1) / d1);
    dq[0] = w1; dq[1] = w2; dq[2] = dd1; dq[3] = dd2;
  }
  static PQN_D float wrap(float x) {
    const float m = -3.14159265358979323846f, M = 3.14159265358979323846f, diff = M - m;
    const int up = x < m, down = x >= M;
    const float how = (float)up * ceilf((m - x) / diff) + (float)down * floorf((x - M) / diff + 1.0f);
    return (x - (how * diff) * (float)down) + (how * diff) * (float)up;
  }
  PQN_D int done_angle() const {
    float s, c1, c12;
    pqn_sincos_f32(y[0], s, c1);
    pqn_sincos_f32(y[1] + y[0], s, c12);
    return (-c1 - c12) > 1.0f;
  }
  PQN_D float step(int action, uint64_t, uint32_t, int &done) {
    const float dt = 0.2f, a = (float)(action - 1);
    float k1[4], k2[4], k3[4], k4[4], q[4];
    dsdt(y, a, k1);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = y[i] + (dt * 0.5f) * k1[i];
    dsdt(q, a, k2);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = y[i] + (dt * 0.5f) * k2[i];
    dsdt(q, a, k3);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = y[i] + dt * k3[i];
    dsdt(q, a, k4);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = y[i] + (dt / 6.0f) * (((k1[i] + 2.0f * k2[i]) + 2.0f * k3[i]) + k4[i]);
    y[0] = wrap(q[0]);
    y[1] = wrap(q[1]);
    y[2] = fminf(fmaxf(q[2], -12.566370614359172f), 12.566370614359172f);
    y[3] = fminf(fmaxf(q[3], -28.274333882308138f), 28.274333882308138f);
    const int da = done_angle();
    time += 1;
    done = da | (time >= MAX_STEPS);
    return -1.0f * (float)(1 - da);
  }
  PQN_D void obs_f32(float *o) const {
    float s, c;
    pqn_sincos_f32(y[0], s, c);
    o[0] = c; o[1] = s;
    pqn_sincos_f32(y[1], s, c);
    o[2] = c; o[3] = s;
    o[4] = y[2]; o[5] = y[3];
  }
  PQN_D void to_canon(int32_t *si, float *sf) const {
    si[0] = time; sf[0] = y[0]; sf[1] = y[1]; sf[2] = y[2]; sf[3] = y[3];
  }
  PQN_D void from_canon(const int32_t *si, const float *sf) {
    time = si[0]; y[0] = sf[0]; y[1] = sf[1]; y[2] = sf[2]; y[3] = sf[3];
  }
};

// ===========================================================================
// LogWrapper record (utils/craftax_wrappers.py:151-200), fused into the step.
// ===========================================================================
struct LogRec {
  float ep_ret;
  int ep_len;
  float ret_ret;
  int ret_len;
  int timestep;
  PQN_D void load(const uint32_t *st, int n, int e, int base) {
    ep_ret = __uint_as_float(st[(size_t)(base + 0) * n + e]);
    ep_len = (int)st[(size_t)(base + 1) * n + e];
    ret_ret = __uint_as_float(st[(size_t)(base + 2) * n + e]);
    ret_len = (int)st[(size_t)(base + 3) * n + e];
    timestep = (int)st[(size_t)(base + 4) * n + e];
  }
  PQN_D void store(uint32_t *st, int n, int e, int base) const {
    st[(size_t)(base + 0) * n + e] = __float_as_uint(ep_ret);
    st[(size_t)(base + 1) * n + e] = (uint32_t)ep_len;
    st[(size_t)(base + 2) * n + e] = __float_as_uint(ret_ret);
    st[(size_t)(base + 3) * n + e] = (uint32_t)ret_len;
    st[(size_t)(base + 4) * n + e] = (uint32_t)timestep;
  }
  PQN_D void zero() { ep_ret = 0.f; ep_len = 0; ret_ret = 0.f; ret_len = 0; timestep = 0; }
  PQN_D void step(float reward, int done) {
    const float new_ret = ep_ret + reward;
    const int new_len = ep_len + 1;
    // the arithmetic form of the reference (utils/craftax_wrappers.py:186-194), not selects: with negative rewards
    // (Craftax) x * 0 is -0.0 and the record must carry the same bits
    const float keep = (float)(1 - done), take = (float)done;
    ep_ret = new_ret * keep;
    ep_len = done ? 0 : new_len;
    ret_ret = ret_ret * keep + new_ret * take;
    ret_len = done ? new_len : ret_len;
    timestep += 1;
  }
};

