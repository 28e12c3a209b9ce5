"""The semi-global recurrence in DIFFERENCE form ("what would move the sweep next"), checked against the
plain recurrence on random relatives -- numpy only, no GPU, no library.

Frame: G(i, j) = H(i, j) + i + j for the reference's H (match +1, mismatch -1, gap -1; source.cpp:1921-1931 without the
X-drop rule: the calm windows' case).  Then
    G(i, j) = max(G(i-1, j-1) + d, G(i-1, j), G(i, j-1)),   d = 3 on a match, 1 otherwise,
and with dH(i, j) = G(i, j) - G(i, j-1), dV(i, j) = G(i, j) - G(i-1, j):
    m = max(d, dH(i-1, j), dV(i, j-1));   dV(i, j) = m - dH(i-1, j);   dH(i, j) = m - dV(i, j-1)
    tag(i, j) = diagonal if d == m, else up if dH(i-1, j) == m, else left          (the reference's order, :1962-1971)
Claims checked: the differences stay in 0 .. 3; the values rebuilt from them equal the plain recurrence's; the tags equal the
plain recurrence's tags.  (Full matrix, no band: the band only restricts which cells exist.)"""
import numpy as np

rng = np.random.default_rng(7)
worst = 0
for trial in range(20):
    n = int(rng.integers(40, 160))
    a = rng.integers(0, 4, n)
    b = np.where(rng.random(n) < rng.random() * 0.5, rng.integers(0, 4, n), a)
    if trial % 3 == 0:
        b = np.roll(b, int(rng.integers(1, 9)))
    # plain recurrence, semi-global start: H(0, 0) = 0, first row / column reachable by gaps only
    H = np.full((n + 1, n + 1), -10**6)
    H[0, :] = -np.arange(n + 1)
    H[:, 0] = -np.arange(n + 1)
    tag = np.zeros((n + 1, n + 1), int)
    for i in range(1, n + 1):
        for j in range(1, n + 1):
            s = 1 if a[i - 1] == b[j - 1] else -1
            cands = (H[i - 1, j - 1] + s, H[i - 1, j] - 1, H[i, j - 1] - 1)          # diagonal, up, left
            H[i, j] = max(cands)
            tag[i, j] = 3 if cands[0] == H[i, j] else 2 if cands[1] == H[i, j] else 1
    G = H + np.add.outer(np.arange(n + 1), np.arange(n + 1))
    # difference form
    dH = np.zeros((n + 1, n + 1), int)
    dV = np.zeros((n + 1, n + 1), int)
    dH[0, 1:] = 0                                           # G(0, j) = 0 for all j: H(0, j) = -j
    dV[1:, 0] = 0
    t2 = np.zeros((n + 1, n + 1), int)
    for i in range(1, n + 1):
        for j in range(1, n + 1):
            d = 3 if a[i - 1] == b[j - 1] else 1
            h, v = dH[i - 1, j], dV[i, j - 1]
            m = max(d, h, v)
            dV[i, j], dH[i, j] = m - h, m - v
            t2[i, j] = 3 if d == m else 2 if h == m else 1
    assert dH.min() >= 0 and dV.min() >= 0 and dH.max() <= 3 and dV.max() <= 3
    worst = max(worst, dH.max(), dV.max())
    G2 = np.zeros((n + 1, n + 1), int)
    G2[1:, 0] = np.cumsum(dV[1:, 0])
    for i in range(n + 1):
        G2[i, 1:] = G2[i, 0] + np.cumsum(dH[i, 1:])
    assert np.array_equal(G2, G), trial
    assert np.array_equal(t2[1:, 1:], tag[1:, 1:]), trial
    # neighbours on an anti-diagonal: G(i-1, j+1) - G(i, j) = dH(i-1, j+1) - dV(i, j)
    for i in range(2, n + 1):
        for j in range(1, n):
            assert G[i - 1, j + 1] - G[i, j] == dH[i - 1, j + 1] - dV[i, j]
print("difference form = plain recurrence on 20 random relatives (values, tags, anti-diagonal neighbours); largest difference seen:", worst)
