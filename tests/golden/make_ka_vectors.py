"""Derives the Q(lambda) known-answer numbers in tests/golden/ka_vectors.json.

Pure-Python float64 transliteration of the two scan bodies of the reference
(minatar form purejaxql/pqn_minatar.py:237-260, atari form
purejaxql/pqn_atari.py:280-302) for a single env.  The reference itself cannot
be imported in the build container (no jax), so these are hand-derived pins,
not reference-generated goldens.  Run: python tests/golden/make_ka_vectors.py
"""


def q_lambda(reward, done, qmax, last_q, gamma, lam, form):
    T = len(reward)
    if form == "minatar":
        last_q = last_q * (1 - done[-1])
        lr = reward[-1] + gamma * last_q
        nq = last_q
    else:
        lr = reward[-1] + gamma * (1 - done[-1]) * last_q
        nq = qmax[-1]
    out = [0.0] * T
    out[-1] = lr
    for t in range(T - 2, -1, -1):
        tb = reward[t] + gamma * (1 - done[t]) * nq
        delta = lr - nq
        lr = tb + gamma * lam * delta
        lr = (1 - done[t]) * lr + done[t] * reward[t]
        nq = qmax[t]
        out[t] = lr
    return out


if __name__ == "__main__":
    r, q, lq = [1, 0, 2, 1], [5, 6, 7, 8], 9
    for d in ([0, 0, 0, 0], [0, 1, 0, 1]):
        for form in ("minatar", "atari"):
            print(d, form, [round(x, 6) for x in q_lambda(r, d, q, lq, 0.99, 0.65, form)])
