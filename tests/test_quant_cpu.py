"""W8A16 quantiser (vz_hip/quant.py) against its restatement in the oracle, on the CPU."""
import torch


def test_row_quantiser_matches_oracle_and_is_bf16_exact():
    from oracle import vz_oracle as O
    from vz_hip import quant
    g = torch.Generator().manual_seed(0)
    w = torch.randn(96, 2048, generator=g) * 0.02
    w[3] *= 500.0
    w[4] = 0.0
    w[5, 7] = 448.0 * 2 ** -12          # exactly on a scale boundary
    w8, scale = quant.quantize_rows(w)
    assert w8.dtype == torch.uint8 and scale.dtype == torch.float32 and w8.shape == w.shape
    e = torch.log2(scale)
    assert torch.equal(e, e.round())                                   # power-of-two scales
    wq = quant.dequantize_rows(w8, scale)
    assert torch.equal(wq, O.fake_quantize_rows(w))                    # same rounding in the oracle
    assert torch.equal(wq.bfloat16().float(), wq)                      # dequantised weights are bf16 numbers
    assert float(w8.view(torch.float8_e4m3fn).float().abs().max()) <= 448.0
    assert float(wq[4].abs().max()) == 0.0 and float(scale[4]) == 1.0
    rel = (wq - w).abs() / w.abs().amax(1, keepdim=True).clamp_min(1e-30)
    assert float(rel.max()) <= 2 ** -4                                 # half a step of a 3-bit mantissa at full scale
    # scales use the full e4m3 range: the largest magnitude of a row lands in (224, 448]
    top = w8.view(torch.float8_e4m3fn).float().abs().amax(1)
    nz = w.abs().amax(1) > 0
    assert bool(((top[nz] > 224) & (top[nz] <= 448)).all())


def test_quantize_state_dict_touches_only_zephyr_linears():
    from oracle import vz_oracle as O
    sd = {"model.layers.0.self_attn.q_proj.weight": torch.randn(8, 64), "model.layers.0.input_layernorm.weight": torch.ones(64),
          "lm_head.weight": torch.randn(10, 64), "model.embed_tokens.weight": torch.randn(10, 64),
          "model.mm_projector.blocks.0.ffn.0.weight": torch.randn(8, 64)}
    q = O.quantize_state_dict(sd)
    for k in sd:
        changed = not torch.equal(q[k], sd[k])
        assert changed == (k in ("model.layers.0.self_attn.q_proj.weight", "lm_head.weight")), k


def test_oracle_fp8_activation_policy_cpu():
    """the oracle's activation quantiser for the fp8 MFMA prefill: Prec(act_fp8=True).q_in is the weights' quantiser applied to the rows of the
    linear's input in prefill, the identity in decode steps and under the other policies; every value it returns is 2^e * e4m3."""
    import torch
    from oracle import vz_oracle as O
    from vz_hip import quant
    torch.manual_seed(0)
    x = torch.randn(2, 5, 64) * 3
    q = O.FP32_FP8ACT.q_in(x, True)
    assert torch.equal(q.reshape(-1, 64), quant.fake_quantize_rows(x.reshape(-1, 64)))
    assert torch.equal(O.FP32_FP8ACT.q_in(x, False), x) and torch.equal(O.BF16.q_in(x, True), x)
    q8, sc = quant.quantize_rows(q.reshape(-1, 64))
    assert torch.equal(quant.dequantize_rows(q8, sc), q.reshape(-1, 64))      # a fixed point: already on the e4m3 grid
