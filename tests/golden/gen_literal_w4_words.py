"""Prints the literal AutoAWQ / AutoGPTQ words of tests/test_w4_layouts.py (LIT_* tables) from the closed forms, with plain
loops that follow the formats' published packing rules -- independent of oracle/w4_layouts.py and of the product's packers.

    python tests/golden/gen_literal_w4_words.py
"""
K, N, G = 16, 16, 8
q = [[(5 * k + 3 * n + (k * n) % 7) % 16 for n in range(N)] for k in range(K)]
z = [[1 + (2 * g + 3 * n) % 15 for n in range(N)] for g in range(K // G)]
ORDER = (0, 2, 4, 6, 1, 3, 5, 7)  # AutoAWQ: nibble i of a word holds column 8j + ORDER[i]


def word(vals8, order=range(8)):
    w = 0
    for i, src in enumerate(order):
        w |= vals8[src] << (4 * i)
    return w


tables = {
    "LIT_AWQ_QW": [[word(q[k][8 * j:8 * j + 8], ORDER) for j in range(N // 8)] for k in range(K)],
    "LIT_AWQ_QZ": [[word(z[g][8 * j:8 * j + 8], ORDER) for j in range(N // 8)] for g in range(K // G)],
    "LIT_GPTQ_QW": [[word([q[8 * r + i][n] for i in range(8)]) for n in range(N)] for r in range(K // 8)],
    "LIT_GPTQ_QZ": [[word([(z[g][8 * j + i] - 1) & 0xF for i in range(8)]) for j in range(N // 8)] for g in range(K // G)],
}
for name, rows in tables.items():
    print(name, "= [" + ",\n    ".join("[" + ", ".join("0x%08X" % w for w in r) + "]" for r in rows) + "]")
