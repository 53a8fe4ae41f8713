"""Not a test: a CPU study of where the bf16 error of the text tower comes from.

It re-runs the oracle's text tower with bf16 rounding inserted at each place the HIP path rounds
(LayerNorm output h, weights w, q/k, v, softmax P, attention output a, MLP hidden g) and prints the
error of the unit-norm embedding against fp32.  Result on the synthetic ViT-B/32 weights (seed 52):
all roundings on -> rms 3.2e-4, max 1.1e-3 over 8 x 512 elements; weights alone account for half of
the variance, layer 0 alone for 42 % (the residual stream starts at the tiny token embeddings).
That is the floor of bf16 x bf16 MFMA operands, independent of kernel quality; the GPU tests
(tests/test_gpu_towers.py) therefore assert 1e-3 on visual features and cosine similarities and
rms 4e-4 / max 1.5e-3 on text features.   Run:  python tests/bf16_floor_study.py
"""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arch as A
from oracle.clip_ref import layer_norm, quick_gelu, causal_mask
torch.set_grad_enabled(False)
a=A.VIT_B32
sd=A.synth_text(a,52)
txt=A.synth_tokens(8,a,54,empty_frac=0.25)
def r(x,on): return x.bfloat16().float() if on else x
def fwd(flags):
    W=a.transformer_width; heads=a.transformer_heads
    x=sd["token_embedding.weight"][txt]+sd["positional_embedding"]
    mask=causal_mask(txt.shape[1],x.dtype)
    for i in range(a.transformer_layers):
        p=f"transformer.resblocks.{i}"
        h=r(layer_norm(x,sd[p+".ln_1.weight"],sd[p+".ln_1.bias"]),flags['h'])
        qkv=h@r(sd[p+".attn.in_proj_weight"],flags['w']).t()+sd[p+".attn.in_proj_bias"]
        q,k,v=qkv.chunk(3,-1)
        q=r(q,flags['qk']);k=r(k,flags['qk']);v=r(v,flags['v'])
        b,L,_=q.shape
        sh=lambda t:t.reshape(b,L,heads,64).transpose(1,2)
        s=(sh(q)*0.125)@sh(k).transpose(-1,-2)+mask
        pr=s.softmax(-1)
        # bf16 P: unnormalised exp rounded (as kernel), approx by rounding normalised
        pr=r(pr,flags['p'])
        at=(pr@sh(v)).transpose(1,2).reshape(b,L,W)
        at=r(at,flags['a'])
        x=x+at@r(sd[p+".attn.out_proj.weight"],flags['w']).t()+sd[p+".attn.out_proj.bias"]
        h=r(layer_norm(x,sd[p+".ln_2.weight"],sd[p+".ln_2.bias"]),flags['h'])
        g=quick_gelu(h@r(sd[p+".mlp.c_fc.weight"],flags['w']).t()+sd[p+".mlp.c_fc.bias"])
        g=r(g,flags['g'])
        x=x+g@r(sd[p+".mlp.c_proj.weight"],flags['w']).t()+sd[p+".mlp.c_proj.bias"]
    x=layer_norm(x,sd["ln_final.weight"],sd["ln_final.bias"])
    o=x[torch.arange(x.shape[0]),txt.argmax(-1)]@sd["text_projection"]
    return o/o.norm(dim=-1,keepdim=True)
keys=['h','w','qk','v','p','a','g']
ref=fwd({k:False for k in keys})
full=fwd({k:True for k in keys})
print("all on: max %.3e rms %.3e"%((full-ref).abs().max(),(full-ref).pow(2).mean().sqrt()))
for k in keys:
    f={kk:(kk==k) for kk in keys}
    o=fwd(f)
    print("only %s: max %.3e rms %.3e"%(k,(o-ref).abs().max(),(o-ref).pow(2).mean().sqrt()))
print("---- per-layer: all roundings on, only in layer i")
import functools
def fwd_layers(active):
    W=a.transformer_width; heads=a.transformer_heads
    x=sd["token_embedding.weight"][txt]+sd["positional_embedding"]
    mask=causal_mask(txt.shape[1],x.dtype)
    for i in range(a.transformer_layers):
        on = i in active
        p=f"transformer.resblocks.{i}"
        h=r(layer_norm(x,sd[p+".ln_1.weight"],sd[p+".ln_1.bias"]),on)
        qkv=h@r(sd[p+".attn.in_proj_weight"],on).t()+sd[p+".attn.in_proj_bias"]
        q,k,v=qkv.chunk(3,-1)
        q=r(q,on);k=r(k,on);v=r(v,on)
        b,L,_=q.shape
        sh=lambda t:t.reshape(b,L,heads,64).transpose(1,2)
        s=(sh(q)*0.125)@sh(k).transpose(-1,-2)+mask
        pr=r(s.softmax(-1),on)
        at=r((pr@sh(v)).transpose(1,2).reshape(b,L,W),on)
        x=x+at@r(sd[p+".attn.out_proj.weight"],on).t()+sd[p+".attn.out_proj.bias"]
        h=r(layer_norm(x,sd[p+".ln_2.weight"],sd[p+".ln_2.bias"]),on)
        g=r(quick_gelu(h@r(sd[p+".mlp.c_fc.weight"],on).t()+sd[p+".mlp.c_fc.bias"]),on)
        x=x+g@r(sd[p+".mlp.c_proj.weight"],on).t()+sd[p+".mlp.c_proj.bias"]
    x=layer_norm(x,sd["ln_final.weight"],sd["ln_final.bias"])
    o=x[torch.arange(x.shape[0]),txt.argmax(-1)]@sd["text_projection"]
    return o/o.norm(dim=-1,keepdim=True)
for i in range(12):
    o=fwd_layers({i})
    print("layer %d: rms %.3e"%(i,(o-ref).pow(2).mean().sqrt()))
o=fwd_layers(set(range(10)))
print("layers 0-9 only: max %.3e rms %.3e"%((o-ref).abs().max(),(o-ref).pow(2).mean().sqrt()))
