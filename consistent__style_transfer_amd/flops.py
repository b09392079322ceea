"""Algorithmic FLOP per sentence of the three stage steps: 2*M*N*K per dense
contraction, forward only, x3 for trained modules (forward + dgrad + wgrad), x2 for frozen critics that only
pass gradient through (forward + dgrad), gathers count 0.  Used by bench.py to turn sentences/s into model
TFLOP/s per stage and pinned against the survey's table in tests/test_host_cpu.py.

Module constants are the reference's (rnn.py:10-12, classifier.py:8-10, discriminator.py:9-12, dim_feedforward 2048)."""

G_EMBED, G_ENC, G_DEC = 128, 256, 512          # generator: token embedding, encoder hidden per direction, decoder hidden
C_EMBED, C_FILTERS, C_KERNELS = 128, 128, (3, 4, 5)
D_EMBED, D_REP, D_FILTERS, D_KERNELS = 128, 16, 300, (2, 3, 4, 5)
D_FF = 2048


def encoder_layer_per_token(d, S):
    return 2 * (4 * d * d + 2 * d * D_FF) + 4 * S * d


def mlm(n_layer, d, L, V):
    return L * (n_layer * encoder_layer_per_token(d, L) + 2 * d * V)


def matcher(n_layer, d, L1, L2):
    S = L1 + L2
    return S * n_layer * encoder_layer_per_token(d, S)


def textcnn(L):
    return sum(2 * (L + k - 1) * k * C_EMBED * C_FILTERS for k in C_KERNELS)


def relgan_d(L):
    es, feat = D_EMBED // D_REP, D_FILTERS * len(D_KERNELS)
    conv = sum(2 * (L - k + 1) * k * es * D_FILTERS for k in D_KERNELS)
    return D_REP * (conv + 2 * feat * feat + 2 * feat * 100 + 2 * 100)


def generator(L_in, T, V):
    """encode L_in positions (both directions), decode T steps (teacher-forced, free-running and soft decoding cost the same)."""
    enc = L_in * 2 * 2 * (G_EMBED + G_ENC) * 4 * G_ENC
    step = 2 * (G_EMBED + G_DEC) * 4 * G_DEC + 4 * L_in * G_DEC + 2 * (G_DEC + 2 * G_ENC) * G_DEC + 2 * G_DEC * V
    return enc + T * step + 2 * 2 * G_ENC * G_DEC            # + transfer


def soft_embed(L, V, width):
    return L * 2 * V * width


def stage_gflop_per_sentence(n_layer, d_model, L, V):
    pre = 3 * (textcnn(L) + matcher(n_layer, d_model, L, L) + mlm(n_layer, d_model, L, V))
    warm = 3 * generator(L, L, V)
    opt_g = (3 * generator(L, L, V) + 3 * generator(L, L, V)                                  # soft decode + back-translation
             + 2 * (matcher(n_layer, d_model, L, L) + soft_embed(L, V, d_model))               # frozen critics: fwd + dgrad
             + 2 * (textcnn(L) + soft_embed(L, V, C_EMBED))
             + 2 * (relgan_d(L) + soft_embed(L, V, D_EMBED)))
    opt_d = generator(L, L, V) + 3 * relgan_d(L) + 3 * (relgan_d(L) + soft_embed(L, V, D_EMBED))   # no-grad decode, D(real ids), D(fake)
    g = 1e-9
    return {"pretrain": pre * g, "warmup": warm * g, "optimize_g": opt_g * g, "optimize_d": opt_d * g}
