"""Where inside ONE Phi-3 layer does the HIP path leave HF's bf16 modules?  Every sub-module of HF's layer 0 (live, on this GPU) is fed to
the matching HIP primitive TEACHER-FORCED (HF's own input of that sub-module) and the outputs are compared."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from transformers import Phi3Config as HFPhi3Config, Phi3ForCausalLM
from dynam3d_amd import dense_ops as D
from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
from dynam3d_amd.weights import synth_state_dict
D.enable_hip_kernels(["all"])
attn_impl = sys.argv[1] if len(sys.argv) > 1 else "sdpa"
c = Phi3Config(layers=1)
bf = torch.bfloat16
sd = synth_state_dict(phi3_param_spec(c), seed=3, device="cuda", dtype_for=lambda n: bf)
with torch.device("meta"):
    hf = Phi3ForCausalLM(HFPhi3Config(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                                      num_key_value_heads=c.kv_heads, rms_norm_eps=c.rms_eps, rope_theta=c.rope_theta, max_position_embeddings=c.max_pos,
                                      original_max_position_embeddings=c.max_pos, pad_token_id=0, tie_word_embeddings=False, attn_implementation=attn_impl)).eval()
hf.load_state_dict({k[len("language_model."):]: v for k, v in sd.items()}, strict=False, assign=True)
inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32, device="cuda") / c.head_dim))
hf.model.rotary_emb.inv_freq = inv
dec = Phi3Decoder(sd, c, bf, "cuda")
S = 896
x = (torch.randn(1, S, c.hidden, device="cuda") * 0.5).to(bf)
rec = {}
def hook(name):
    def f(m, args, kwargs, out):
        rec[name] = (args, kwargs, out)
    return f
lay = hf.model.layers[0]
hs = [lay.input_layernorm.register_forward_hook(hook("n1"), with_kwargs=True), lay.self_attn.qkv_proj.register_forward_hook(hook("qkv"), with_kwargs=True),
      lay.self_attn.o_proj.register_forward_hook(hook("o"), with_kwargs=True), lay.post_attention_layernorm.register_forward_hook(hook("n2"), with_kwargs=True),
      lay.mlp.gate_up_proj.register_forward_hook(hook("gu"), with_kwargs=True), lay.mlp.down_proj.register_forward_hook(hook("down"), with_kwargs=True),
      lay.mlp.register_forward_hook(hook("mlp"), with_kwargs=True), lay.self_attn.register_forward_hook(hook("attn"), with_kwargs=True),
      lay.register_forward_hook(hook("layer"), with_kwargs=True)]
with torch.no_grad():
    hf(inputs_embeds=x)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
L = dec.layers[0]
Tp = (S + 255) // 256 * 256
def pad(t):
    o = torch.zeros((Tp, t.shape[-1]), dtype=bf, device="cuda"); o[:S] = t.reshape(S, -1); return o
ctx = dec.packed_context([S], Tp)
print("attention implementation of the HF reference:", attn_impl)
# RMSNorm
n1_in, n1_out = rec["n1"][0][0][0], rec["n1"][2][0]
print("input_layernorm      ", rel(D.rms_norm(pad(n1_in), L["n1"], c.rms_eps)[:S], n1_out))
qkv_in, qkv_out = rec["qkv"][0][0][0], rec["qkv"][2][0]
mine_qkv = D.linear(pad(qkv_in), L["qkv_w"], None)
print("qkv_proj             ", rel(mine_qkv[:S], qkv_out), " mismatching elements: %.3f %%" % (100 * float((mine_qkv[:S] != qkv_out).float().mean())))
# rope + attention, teacher-forced on HF's qkv: output = o_proj's input
o_in, o_out = rec["o"][0][0][0], rec["o"][2][0]
q2 = pad(qkv_out).clone()
D.rope_packed_(q2, c.heads + c.kv_heads, c.head_dim, ctx["cos"], ctx["sin"], ctx["pos"])
a = D.attention_packed(q2.view(Tp, 3 * c.heads, c.head_dim), c.heads, True, ctx["cu"], 1, S, n_valid=S).view(Tp, -1)
print("rope + attention     ", rel(a[:S], o_in))
# float32 reference of rope + attention on HF's qkv (what both approximate)
qf = qkv_out.float().view(S, 3 * c.heads, c.head_dim)
cos = torch.cat([ctx["cos"], ctx["cos"]], -1)[:S, None]; sin = torch.cat([ctx["sin"], ctx["sin"]], -1)[:S, None]
def rot(t):
    h = t.shape[-1] // 2
    return torch.cat([-t[..., h:], t[..., :h]], -1)
qq, kk, vv = qf[:, :c.heads], qf[:, c.heads:2 * c.heads], qf[:, 2 * c.heads:]
qq = (qq * cos + rot(qq) * sin); kk = (kk * cos + rot(kk) * sin)
ref32 = torch.nn.functional.scaled_dot_product_attention(qq.transpose(0, 1)[None], kk.transpose(0, 1)[None], vv.transpose(0, 1)[None], is_causal=True)[0].transpose(0, 1).reshape(S, -1)
print("   HF  attention vs float32 rope+attention", rel(o_in.float(), ref32), "   HIP vs float32", rel(a[:S].float(), ref32))
# rope only: compare against HF's bf16 formula
qh = qkv_out.view(S, 3 * c.heads, c.head_dim)
cb, sb = cos.to(bf), sin.to(bf)
hf_rope_q = (qh[:, :2 * c.heads] * cb) + (rot(qh[:, :2 * c.heads]) * sb)
print("rope (q,k) vs HF's bf16 formula", rel(q2[:S].view(S, 3 * c.heads, c.head_dim)[:, :2 * c.heads], hf_rope_q), " mismatching: %.3f %%" % (100 * float((q2[:S].view(S, 3 * c.heads, c.head_dim)[:, :2 * c.heads] != hf_rope_q).float().mean())))
# o_proj + residual
res_in = n1_in
mine_o = D.linear(pad(o_in), L["o_w"], None, residual=pad(res_in))
hf_after_attn = (res_in + o_out)
print("o_proj + residual    ", rel(mine_o[:S], hf_after_attn))
n2_in, n2_out = rec["n2"][0][0][0], rec["n2"][2][0]
print("post_attention_norm  ", rel(D.rms_norm(pad(n2_in), L["n2"], c.rms_eps)[:S], n2_out))
gu_in, gu_out = rec["gu"][0][0][0], rec["gu"][2][0]
down_in, down_out = rec["down"][0][0][0], rec["down"][2][0]
act = D.linear_swiglu(pad(gu_in), L["gu_w"], dec.interleave_gu)
print("gate_up + SwiGLU     ", rel(act[:S], down_in), " mismatching: %.3f %%" % (100 * float((act[:S] != down_in).float().mean())))
mine_d = D.linear(pad(down_in), L["down_w"], None, residual=pad(n2_in))
print("down_proj + residual ", rel(mine_d[:S], (n2_in + down_out)))
lay_out = rec["layer"][2]; lay_out = lay_out[0] if isinstance(lay_out, (tuple, list)) else lay_out
print("whole layer          ", rel(dec.layer_packed(0, pad(x[0]), ctx)[:S], lay_out[0]))
