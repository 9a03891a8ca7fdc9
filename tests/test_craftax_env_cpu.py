"""Hand-derived known-answer steps for the Craftax-Classic restatement (oracle/craftax_classic.c): each case builds a
canonical state by hand, applies actions and checks the outcome the published Crafter / Craftax-Classic rules give
(data.yaml collect / place / make tables, Player.update vitals, Zombie attack, observation layout).  These pin the
oracle's RULES against the documented mechanics; the HIP kernels are pinned to the oracle in test_craftax_env_gpu.py.
(The env itself is third-party and not under /root/reference: gymnax-style parity with the real craftax package stays
unpinned, see the header of oracle/craftax_classic.c.)"""
import numpy as np
import pytest

S = 4096
GRASS, WATER, STONE, TREE, PATH, COAL, IRON, DIAMOND, TABLE, FURNACE, SAND, LAVA, PLANT, RIPE = 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16
NOOP, LEFT, RIGHT, UP, DOWN, DO, SLEEP, P_STONE, P_TABLE, P_FURNACE, P_PLANT, M_WPICK, M_SPICK, M_IPICK, M_WSWORD, M_SSWORD, M_ISWORD = range(17)
NAME = "Craftax-Classic-Symbolic-v1"


def fresh(oracle, n=1):
    env = oracle.OracleEnv(NAME)
    _obs, st = env.reset(1, n)
    st["si"][:, :S] = GRASS            # a meadow: every case places its own blocks
    return env, st


def put(st, r, c, b, e=0):
    st["si"][e, r * 64 + c] = b


def cell(st, r, c, e=0):
    return int(st["si"][e, r * 64 + c])


def sc(st, i, e=0):
    return int(st["si"][e, S + i])


def act(env, st, a, key=0):
    obs, st, r, d, info = env.step(key, st, np.asarray([a], np.int32), autoreset=False)
    return obs[0], float(r[0]), bool(d[0])


def test_spec_and_reset(oracle):
    env = oracle.OracleEnv(NAME)
    assert env.obs_shape == (1345,) and env.num_actions == 17 and env.max_steps == 10000
    obs, st = env.reset(3, 4)
    s = st["si"][:, S:]
    assert (s[:, 0] == 32).all() and (s[:, 1] == 32).all() and (s[:, 3:7] == 9).all() and (s[:, 8:20] == 0).all()
    assert (st["si"][:, 32 * 64 + 32] == GRASS).all()                  # the spawn is always a clearing
    assert len({st["si"][e, :S].tobytes() for e in range(4)}) == 4     # every env has its own world
    assert set(np.unique(st["si"][:, :S])) <= {GRASS, WATER, STONE, TREE, PATH, COAL, IRON, DIAMOND, SAND, LAVA}
    np.testing.assert_array_equal(obs.sum(1) > 60, True)


def test_collect_place_make_chain(oracle):
    """wood -> table -> wood pickaxe -> stone -> stone pickaxe; every first-time achievement pays +1."""
    env, st = fresh(oracle)
    put(st, 33, 32, TREE)                                   # the player faces down at reset
    st["si"][0, S + 4] = 5                                  # food 5: no health regeneration effects on the reward
    for k in range(3):
        _, r, d = act(env, st, DO)
        assert r == (1.0 if k == 0 else 0.0) and not d      # COLLECT_WOOD once
    assert sc(st, 8) == 3 and cell(st, 33, 32) == TREE      # the tree stays
    _, r, _ = act(env, st, LEFT)                            # turn / step left onto (32, 31)
    assert (sc(st, 0), sc(st, 1), sc(st, 2)) == (32, 31, LEFT) and r == 0.0
    _, r, _ = act(env, st, P_TABLE)                         # table on (32, 30), costs 1 wood
    assert r == 1.0 and cell(st, 32, 30) == TABLE and sc(st, 8) == 2
    _, r, _ = act(env, st, M_WPICK)
    assert r == 1.0 and sc(st, 8 + 6) == 1 and sc(st, 8) == 1
    _, r, _ = act(env, st, M_SPICK)                         # no stone yet
    assert r == 0.0 and sc(st, 8 + 7) == 0
    put(st, 31, 31, STONE)
    act(env, st, UP)                                        # blocked by the stone, but now facing it
    assert (sc(st, 0), sc(st, 1), sc(st, 2)) == (32, 31, UP)
    _, r, _ = act(env, st, DO)
    assert r == 1.0 and sc(st, 8 + 1) == 1 and cell(st, 31, 31) == PATH
    _, r, _ = act(env, st, M_SPICK)                         # table is adjacent: wood 1 + stone 1
    assert r == 1.0 and sc(st, 8 + 7) == 1 and sc(st, 8) == 0 and sc(st, 8 + 1) == 0
    assert bin(sc(st, 109)).count("1") == 5


def test_pickaxe_requirements_and_iron_tools(oracle):
    env, st = fresh(oracle)
    put(st, 33, 32, IRON)
    _, r, _ = act(env, st, DO)
    assert r == 0.0 and cell(st, 33, 32) == IRON                       # iron needs a stone pickaxe
    st["si"][0, S + 8 + 7] = 1
    _, r, _ = act(env, st, DO)
    assert r == 1.0 and sc(st, 8 + 3) == 1 and cell(st, 33, 32) == PATH
    st["si"][0, S + 8:S + 12] = [1, 1, 1, 1]                           # wood, stone, coal, iron
    put(st, 32, 31, TABLE)
    _, r, _ = act(env, st, M_IPICK)
    assert r == 0.0                                                    # no furnace nearby
    put(st, 31, 32, FURNACE)
    _, r, _ = act(env, st, M_IPICK)
    assert r == 1.0 and sc(st, 8 + 8) == 1 and [sc(st, 8), sc(st, 10), sc(st, 11)] == [0, 0, 0] and sc(st, 9) == 1
    put(st, 33, 32, DIAMOND)
    _, r, _ = act(env, st, DO)
    assert r == 1.0 and sc(st, 8 + 4) == 1


def test_water_lava_and_bounds(oracle):
    env, st = fresh(oracle)
    st["si"][0, S + 5] = 3
    put(st, 33, 32, WATER)
    _, r, _ = act(env, st, DO)
    assert r == 1.0 and sc(st, 5) == 4                                 # COLLECT_DRINK
    act(env, st, DOWN)
    assert sc(st, 0) == 32                                             # water is not walkable
    put(st, 32, 33, LAVA)
    _, r, d = act(env, st, RIGHT)
    assert d and (sc(st, 0), sc(st, 1)) == (32, 33)                    # lava is enterable and deadly
    env, st = fresh(oracle)
    st["si"][0, S + 0], st["si"][0, S + 1] = 0, 0
    _, _, d = act(env, st, UP)
    assert (sc(st, 0), sc(st, 1), sc(st, 2)) == (0, 0, UP) and not d   # the map edge blocks, the facing turns
    obs, _, _ = act(env, st, NOOP)
    view = obs[:1323].reshape(7, 9, 21)
    assert view[0, 0, 1] == 1.0 and view[3, 4, GRASS] == 1.0 and view[2, 4, 1] == 1.0   # out-of-bounds cells around the corner


def test_vitals_clock(oracle):
    """Player.update: food drops after 26 awake steps, drink after 21, energy after 31; health regenerates every 26."""
    env, st = fresh(oracle)
    st["si"][0, S + 3] = 5
    food, drink, energy, health = [], [], [], []
    for t in range(32):
        act(env, st, NOOP, key=t)
        food.append(sc(st, 4)); drink.append(sc(st, 5)); energy.append(sc(st, 6)); health.append(sc(st, 3))
    assert food[24] == 9 and food[25] == 8 and drink[19] == 9 and drink[20] == 8 and energy[29] == 9 and energy[30] == 8
    assert health[24] == 5 and health[25] == 6


def test_zombie_attack_and_defeat(oracle):
    env, st = fresh(oracle)
    st["si"][0, S + 20:S + 25] = [33, 32, 5, 0, 1]                     # a zombie right below, cooldown 0
    st["si"][0, S + 35:S + 47] = 0
    _, r, _ = act(env, st, NOOP, key=11)
    assert sc(st, 3) == 7 and abs(r - (-0.2)) < 1e-6 and sc(st, 23) == 5            # 2 damage, cooldown 5
    st["si"][0, S + 8 + 10] = 1                                        # stone sword: 3 damage per hit
    if (sc(st, 20), sc(st, 21)) == (33, 32):
        act(env, st, DO, key=12)
        assert sc(st, 22) == 2
        if (sc(st, 20), sc(st, 21)) == (33, 32):
            _, r, _ = act(env, st, DO, key=13)
            assert sc(st, 24) == 0 and sc(st, 109) & (1 << 8)                       # DEFEAT_ZOMBIE


def test_sleep_and_wake(oracle):
    env, st = fresh(oracle)
    st["si"][0, S + 6] = 8
    obs, r, _ = act(env, st, SLEEP)
    assert sc(st, 7) == 1 and obs[1344] == 1.0
    pos = (sc(st, 0), sc(st, 1))
    woke = None
    for t in range(40):
        obs, r, _ = act(env, st, LEFT, key=100 + t)                    # ignored while asleep
        if sc(st, 7) == 0:
            woke = (t, r)
            break
        assert (sc(st, 0), sc(st, 1)) == pos
    assert woke is not None and woke[0] == 9 and woke[1] >= 1.0 and sc(st, 6) == 9 and sc(st, 109) & (1 << 21)   # fatigue -11 after 11 sleeping steps


def test_plant_cycle_and_time_limit(oracle):
    env, st = fresh(oracle)
    st["si"][0, S + 8 + 5] = 1
    _, r, _ = act(env, st, P_PLANT)
    assert r == 1.0 and cell(st, 33, 32) == PLANT and sc(st, 8 + 5) == 0 and sc(st, 69 + 3) == 1
    st["si"][0, S + 69 + 2] = 299
    act(env, st, NOOP, key=1)
    assert cell(st, 33, 32) == PLANT
    act(env, st, NOOP, key=2)
    assert cell(st, 33, 32) == RIPE
    st["si"][0, S + 4] = 3
    _, r, _ = act(env, st, DO, key=3)
    assert r == 1.0 and sc(st, 4) == 7 and cell(st, 33, 32) == PLANT and sc(st, 69 + 2) <= 1
    st["si"][0, S + 110] = 9999
    _, _, d = act(env, st, NOOP, key=4)
    assert d and sc(st, 110) == 10000


def test_observation_layout(oracle):
    env, st = fresh(oracle)
    put(st, 32, 36, TREE); put(st, 29, 28, WATER)
    st["si"][0, S + 8:S + 20] = np.arange(12) % 10
    st["si"][0, S + 35:S + 39] = [31, 32, 3, 1]                        # a cow above the player
    st["si"][0, S + 57:S + 61] = [35, 32, UP, 1]                       # an arrow three cells below
    obs = np.zeros((1, 1345), np.float32)
    oracle.lib().pqn_oracle_env_obs(env.env_id, 1, st["si"].ctypes.data, st["sf"].ctypes.data, obs.ctypes.data)
    v = obs[0, :1323].reshape(7, 9, 21)
    assert v[3, 8, TREE] == 1.0 and v[0, 0, WATER] == 1.0 and v[2, 4, 18] == 1.0 and v[6, 4, 20] == 1.0
    assert v.sum() == 63 + 2                                            # one block per cell + the two mob channels
    t = obs[0, 1323:]
    np.testing.assert_allclose(t[:12], (np.arange(12) % 10) / 10.0, rtol=0, atol=1e-7)
    np.testing.assert_allclose(t[12:16], 0.9, atol=1e-7)
    assert list(t[16:20]) == [0, 0, 0, 1] and t[21] == 0.0 and 0.0 <= t[20] <= 1.0


def test_sword_damage_table_and_eat_cow(oracle):
    """Crafter's damage table (1 bare-handed, 2 wood, 3 stone, 5 iron sword: the best sword held counts) on a cow of 3
    health, and what eating it pays: food + 6 capped at 9, hunger clock cleared, EAT_COW once."""
    for sword_slot, hits in ((None, 3), (9, 2), (10, 1), (11, 1)):          # inventory slots 9 / 10 / 11 = wood / stone / iron sword
        env, st = fresh(oracle)
        st["si"][0, S + 20:S + 35] = 0                                       # no zombies
        st["si"][0, S + 35:S + 47] = 0
        st["si"][0, S + 35:S + 39] = [33, 32, 3, 1]                          # a cow right below (r, c, health, mask) ...
        for rr, cc in ((34, 32), (33, 31), (33, 33)):
            put(st, rr, cc, STONE)                                           # ... walled in: it cannot wander off
        st["si"][0, S + 4] = 2                                               # food 2
        if sword_slot is not None:
            st["si"][0, S + 8 + sword_slot] = 1
        n = 0
        while sc(st, 38) and n < 6:
            assert (sc(st, 35), sc(st, 36)) == (33, 32)
            act(env, st, DO, key=40 + n)
            n += 1
        assert n == hits, (sword_slot, n)
        assert sc(st, 4) == 8 and sc(st, 109) & (1 << 9)                     # food 2 + 6, EAT_COW
        assert float(st["sf"][0, 1]) <= 1.0                                  # hunger clock restarted (at most this step's tick)


def test_stone_bridges_water_and_lava_and_mobs_block_the_way(oracle):
    """PLACE_STONE is allowed onto water and lava (Crafter data.yaml place.stone.where = grass, sand, path, water, lava);
    a cell occupied by a mob can neither be entered nor built on; the facing direction follows the action regardless."""
    env, st = fresh(oracle)
    st["si"][0, S + 20:S + 69] = 0
    st["si"][0, S + 8 + 1] = 3                                               # 3 stones
    put(st, 33, 32, WATER)
    _, r, _ = act(env, st, P_STONE)
    assert cell(st, 33, 32) == STONE and sc(st, 9) == 2 and r == 1.0         # PLACE_STONE achievement
    act(env, st, RIGHT)                                                      # now at (32, 33), facing right
    assert (sc(st, 0), sc(st, 1), sc(st, 2)) == (32, 33, RIGHT)
    put(st, 32, 34, LAVA)
    act(env, st, P_STONE)
    assert cell(st, 32, 34) == STONE and sc(st, 9) == 1
    # a cow in the way: no move, no placement, but the player turns
    st["si"][0, S + 35:S + 39] = [31, 33, 3, 1]
    act(env, st, UP, key=77)
    if (sc(st, 35), sc(st, 36)) == (31, 33):                                 # (unless this draw moved the cow away)
        assert (sc(st, 0), sc(st, 1), sc(st, 2)) == (32, 33, UP)
        before = cell(st, 31, 33)
        act(env, st, P_STONE, key=78)
        if (sc(st, 35), sc(st, 36)) == (31, 33):
            assert cell(st, 31, 33) == before and sc(st, 9) == 1


def test_sleeping_player_ignores_actions(oracle):
    """While is_sleeping the chosen action is replaced by noop (Crafter Player.update: `if self.sleeping: action = noop`):
    no move, no interaction, until energy is back."""
    env, st = fresh(oracle)
    st["si"][0, S + 20:S + 69] = 0
    st["si"][0, S + 6] = 3                                                   # energy 3
    put(st, 33, 32, TREE)
    act(env, st, SLEEP)
    assert sc(st, 7) == 1
    pos = (sc(st, 0), sc(st, 1))
    act(env, st, LEFT, key=5)
    act(env, st, DO, key=6)
    assert sc(st, 7) == 1 and (sc(st, 0), sc(st, 1)) == pos and sc(st, 8) == 0     # still asleep, not moved, no wood collected
