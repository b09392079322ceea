"""Test configurations shared by tests/golden/make_golden.py and the parity tests (TEST INFRASTRUCTURE)."""
CONFIGS = {
    # everything small enough that a pure fp32 oracle run takes milliseconds
    "tiny": dict(V=53, B=3, L=6, max_len=7, d_model=32, n_head=4, n_layer=2,
                 g_embed=16, g_enc=16, g_dec=32, c_embed=16, c_filters=[8, 8, 8],
                 d_embed=32, d_rep=4, d_filters=[12, 12, 12, 12]),
    # the reference's own module constants, small batch / vocabulary
    "ref": dict(V=211, B=2, L=7, max_len=8, d_model=512, n_head=8, n_layer=6,
                g_embed=128, g_enc=256, g_dec=512, c_embed=128, c_filters=[128, 128, 128],
                d_embed=128, d_rep=16, d_filters=[300, 300, 300, 300]),
}


