"""TEST INFRASTRUCTURE ONLY -- torch-CPU restatement of the reference UNet.

Functional form over a plain ``{name: tensor}`` dict that uses the reference's
state-dict key layout (DD:272-361), so a fixture of weights can be fed to the
reference module, to this oracle and to the HIP engine alike.

Two numerics modes:
  * ``mode="fp32"``  : the reference at ``precision: 32`` -- eps 1e-5 everywhere
                       (DD:107, DD:122 with fp32 activations), no rounding.
  * ``mode="autocast"``: no explicit rounding, eps chosen from the live dtype exactly as the
                       reference does; meant to run inside ``torch.autocast`` (CPU policy
                       here -- pinned to the reference's CPU-autocast golden; CUDA policy on
                       the GPU box, where it validates ``site_eps``).
  * ``mode="bf16c"`` : the ENGINE CONTRACT -- fp32 arithmetic on bf16-rounded
                       tensors.  ``q()`` marks every point where the HIP engine
                       stores an activation / weight as bf16, and the per-site
                       eps follows the dtype the reference sees at that site
                       under bf16 autocast (``site_eps``; derived in DESIGN.md,
                       checked against a trace of the reference in
                       tests/test_oracle_unet.py).
"""
import math

import torch
import torch.nn.functional as F

HEADS = 4
DIM_HEAD = 32


def q_bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def q_id(t):
    return t


# --------------------------------------------------------------------------- eps by site
def site_eps(dim_mults=(1, 2, 4, 8)):
    """eps the reference uses at each WS-conv / LayerNorm site under bf16 autocast.

    DD:107 / DD:122: ``eps = 1e-5 if x.dtype == float32 else 1e-3``.  Under autocast conv
    outputs are bf16, GroupNorm/softmax outputs are fp32, ``fp32 + bf16`` and ``cat(bf16,
    fp32)`` promote to fp32.  So a ResnetBlock's block1 sees bf16 only when its input is a
    bare conv output (init_conv, a Downsample/plain conv, cat(up-conv, init_conv)).
    """
    n = len(dim_mults)
    e = {}
    for i in range(n):
        e[f"downs.{i}.0.block1.proj"] = 1e-3   # input: init_conv / downsample conv output (bf16)
        e[f"downs.{i}.0.block2.proj"] = 1e-5
        e[f"downs.{i}.1.block1.proj"] = 1e-5   # input: previous ResnetBlock output (fp32)
        e[f"downs.{i}.1.block2.proj"] = 1e-5
        e[f"downs.{i}.2.fn.norm"] = 1e-5       # PreNorm on fp32
        e[f"downs.{i}.2.fn.fn.to_out.1"] = 1e-3  # LayerNorm on a conv output (bf16)
        e[f"ups.{i}.0.block1.proj"] = 1e-5     # cat(bf16, fp32) -> fp32
        e[f"ups.{i}.0.block2.proj"] = 1e-5
        e[f"ups.{i}.1.block1.proj"] = 1e-5
        e[f"ups.{i}.1.block2.proj"] = 1e-5
        e[f"ups.{i}.2.fn.norm"] = 1e-5
        e[f"ups.{i}.2.fn.fn.to_out.1"] = 1e-3
    e["mid_block1.block1.proj"] = 1e-3         # input: downs[-1] plain conv (bf16)
    e["mid_block1.block2.proj"] = 1e-5
    e["mid_attn.fn.norm"] = 1e-5
    e["mid_block2.block1.proj"] = 1e-5         # conv(bf16) + x(fp32) -> fp32
    e["mid_block2.block2.proj"] = 1e-5
    e["final_res_block.block1.proj"] = 1e-3    # cat(up conv bf16, init_conv clone bf16)
    e["final_res_block.block2.proj"] = 1e-5
    return e


class _Eps:
    """eps chooser.  fp32: 1e-5.  bf16c: the static CUDA-autocast table above.  autocast: the
    reference's own rule on the live dtype (DD:107, DD:122), recorded in ``trace``."""

    def __init__(self, mode, dim_mults):
        self.mode = mode
        self.table = site_eps(dim_mults) if mode == "bf16c" else None
        self.trace = {}

    def __call__(self, site, x):
        if self.mode == "autocast":
            e = 1e-5 if x.dtype == torch.float32 else 1e-3
            self.trace[site] = e
            return e
        return 1e-5 if self.table is None else self.table[site]


# --------------------------------------------------------------------------- blocks
def standardize_weight(w, eps):
    """DD:109-112: per-out-channel (w - mean) * rsqrt(biased var + eps)."""
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return (w - mean) * (var + eps).rsqrt()


def layer_norm_c(x, g, eps):
    """DD:121-125: normalise over the channel dim, gain only."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * (var + eps).rsqrt() * g


def sinusoidal_pos_emb(t, dim):
    """DD:144-151."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -k)
    arg = t.to(torch.float32)[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_mlp(P, t, dim):
    """DD:319-324: SinusoidalPosEmb -> Linear -> GELU(erf) -> Linear."""
    e = sinusoidal_pos_emb(t, dim)
    e = F.linear(e, P["time_mlp.1.weight"], P["time_mlp.1.bias"])
    e = F.gelu(e)
    return F.linear(e, P["time_mlp.3.weight"], P["time_mlp.3.bias"])


def resnet_block(P, pre, x, temb, eps, q, groups=8):
    """DD:190-214 (ResnetBlock) with DD:172-188 (Block).

    x arrives already stored (q'd).  Rounding points of the engine: conv outputs, the
    conv2 input SiLU(GN(h1)), and the block output.
    """
    timed = temb is not None and f"{pre}.mlp.1.weight" in P                        # DD:204 exists(self.mlp) and exists(time_emb)
    if timed:
        ss = F.linear(F.silu(temb), P[f"{pre}.mlp.1.weight"], P[f"{pre}.mlp.1.bias"])  # DD:205-208
        scale, shift = ss[:, :, None, None].chunk(2, dim=1)

    w1 = q(standardize_weight(P[f"{pre}.block1.proj.weight"], eps(f"{pre}.block1.proj", x)))
    h = q(F.conv2d(x, w1, P[f"{pre}.block1.proj.bias"], padding=1))
    h = F.group_norm(h, groups, P[f"{pre}.block1.norm.weight"], P[f"{pre}.block1.norm.bias"], eps=1e-5)
    h = q(F.silu(h * (scale + 1) + shift if timed else h))                         # DD:183-187

    w2 = q(standardize_weight(P[f"{pre}.block2.proj.weight"], eps(f"{pre}.block2.proj", h)))
    h = q(F.conv2d(h, w2, P[f"{pre}.block2.proj.bias"], padding=1))
    h = F.group_norm(h, groups, P[f"{pre}.block2.norm.weight"], P[f"{pre}.block2.norm.bias"], eps=1e-5)
    h = F.silu(h)

    if f"{pre}.res_conv.weight" in P:                                              # DD:200
        res = F.conv2d(x, q(P[f"{pre}.res_conv.weight"]), P[f"{pre}.res_conv.bias"])
    else:
        res = x
    return q(h + res)                                                              # DD:214


def linear_attention(P, pre, x, eps, q):
    """DD:81-87 Residual(DD:127-135 PreNorm(DD:216-244 LinearAttention)); pre = '<..>.2'."""
    b, c, h, w = x.shape
    n = h * w
    xn = q(layer_norm_c(x, P[f"{pre}.fn.norm.g"], eps(f"{pre}.fn.norm", x)))
    qkv = q(F.conv2d(xn, q(P[f"{pre}.fn.fn.to_qkv.weight"])))
    qq, kk, vv = [t.reshape(b, HEADS, DIM_HEAD, n) for t in qkv.chunk(3, dim=1)]   # DD:232
    qq = qq.softmax(dim=-2) * (DIM_HEAD ** -0.5)                                   # DD:234,237
    kk = kk.softmax(dim=-1)                                                        # DD:235
    vv = vv / n                                                                    # DD:238
    ctx = torch.einsum("bhdn,bhen->bhde", kk, vv)                                  # DD:240
    out = torch.einsum("bhde,bhdn->bhen", ctx, qq)                                 # DD:242
    out = q(out.reshape(b, HEADS * DIM_HEAD, h, w))
    out = q(F.conv2d(out, q(P[f"{pre}.fn.fn.to_out.0.weight"]), P[f"{pre}.fn.fn.to_out.0.bias"]))
    out = layer_norm_c(out, P[f"{pre}.fn.fn.to_out.1.g"], eps(f"{pre}.fn.fn.to_out.1", out))
    return q(out + x)


def attention(P, pre, x, eps, q):
    """Residual(PreNorm(DD:246-268 Attention)); pre = 'mid_attn'."""
    b, c, h, w = x.shape
    n = h * w
    xn = q(layer_norm_c(x, P[f"{pre}.fn.norm.g"], eps(f"{pre}.fn.norm", x)))
    qkv = q(F.conv2d(xn, q(P[f"{pre}.fn.fn.to_qkv.weight"])))
    qq, kk, vv = [t.reshape(b, HEADS, DIM_HEAD, n) for t in qkv.chunk(3, dim=1)]
    qq = qq * (DIM_HEAD ** -0.5)                                                   # DD:261
    sim = torch.einsum("bhdi,bhdj->bhij", qq, kk)                                  # DD:263
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, vv)                                # DD:265
    out = q(out.permute(0, 1, 3, 2).reshape(b, HEADS * DIM_HEAD, h, w))            # DD:267
    out = F.conv2d(out, q(P[f"{pre}.fn.fn.to_out.weight"]), P[f"{pre}.fn.fn.to_out.bias"])
    return q(out + x)


def downsample(P, pre, x, q):
    """DD:95-99: 'b c (h p1) (w p2) -> b (c p1 p2) h w' then 1x1 conv."""
    b, c, h, w = x.shape
    x = x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, h // 2, w // 2)
    return q(F.conv2d(x, q(P[f"{pre}.1.weight"]), P[f"{pre}.1.bias"]))


def upsample(P, pre, x, q):
    """DD:89-93: nearest x2 then 3x3 conv."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return q(F.conv2d(x, q(P[f"{pre}.1.weight"]), P[f"{pre}.1.bias"], padding=1))


def unet_forward(P, x, cond, t, dim=64, dim_mults=(1, 2, 4, 8), mode="fp32", taps=None, eps_trace=None, eps_table=None):
    """DD:363-417.  x (B,Cx,H,W), cond (B,Cc,H,W) or None, t (B,) int64.  Returns (B,out_dim,H,W).

    ``taps``: optional dict that receives named intermediates (NCHW fp32) for per-stage checks.
    """
    q = q_bf16 if mode == "bf16c" else q_id
    eps = _Eps(mode, dim_mults)
    if eps_table is not None:       # e.g. the CPU-autocast trace, to compare against that golden
        eps.table = dict(eps_table)
    n = len(dim_mults)

    def tap(name, v):
        if taps is not None:
            taps[name] = v.clone()
        return v

    if cond is not None:
        x = torch.cat((x, cond), dim=1)                                            # DD:368
    x = q(x)
    x = q(F.conv2d(x, q(P["init_conv.weight"]), P["init_conv.bias"], padding=3))   # DD:374
    r = x
    tap("init_conv", x)
    temb = None                                                                    # DD:376-385: time_in=False takes no time
    if "time_mlp.1.weight" in P:
        temb = time_mlp(P, t, dim)                                                 # DD:381
        tap("temb", temb)

    hs = []
    for i in range(n):                                                             # DD:390-398
        x = resnet_block(P, f"downs.{i}.0", x, temb, eps, q)
        hs.append(x)
        tap(f"downs.{i}.0", x)
        x = resnet_block(P, f"downs.{i}.1", x, temb, eps, q)
        x = linear_attention(P, f"downs.{i}.2", x, eps, q)
        hs.append(x)
        tap(f"downs.{i}.2", x)
        if i < n - 1:
            x = downsample(P, f"downs.{i}.3", x, q)
        else:
            x = q(F.conv2d(x, q(P[f"downs.{i}.3.weight"]), P[f"downs.{i}.3.bias"], padding=1))
        tap(f"downs.{i}.3", x)

    x = resnet_block(P, "mid_block1", x, temb, eps, q)                             # DD:400-402
    tap("mid_block1", x)
    x = attention(P, "mid_attn", x, eps, q)
    tap("mid_attn", x)
    x = resnet_block(P, "mid_block2", x, temb, eps, q)
    tap("mid_block2", x)

    for i in range(n):                                                             # DD:404-412
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(P, f"ups.{i}.0", x, temb, eps, q)
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(P, f"ups.{i}.1", x, temb, eps, q)
        x = linear_attention(P, f"ups.{i}.2", x, eps, q)
        tap(f"ups.{i}.2", x)
        if i < n - 1:
            x = upsample(P, f"ups.{i}.3", x, q)
        else:
            x = q(F.conv2d(x, q(P[f"ups.{i}.3.weight"]), P[f"ups.{i}.3.bias"], padding=1))
        tap(f"ups.{i}.3", x)

    x = torch.cat((x, r), dim=1)                                                   # DD:414
    x = resnet_block(P, "final_res_block", x, temb, eps, q)
    tap("final_res_block", x)
    if eps_trace is not None:
        eps_trace.update(eps.trace)
    return F.conv2d(x, P["final_conv.weight"], P["final_conv.bias"])               # DD:417


# --------------------------------------------------------------------------- weights
def unet_param_shapes(dim=64, channels=5, out_dim=2, dim_mults=(1, 2, 4, 8), time_in=True):
    """Ordered {name: shape} of the reference ``Unet(dim, channels=, out_dim=, time_in=)`` state dict (DD:272-361)."""
    S = {}
    tdim = dim * 4
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)

    def conv(name, co, ci, k, bias=True):
        S[f"{name}.weight"] = (co, ci, k, k)
        if bias:
            S[f"{name}.bias"] = (co,)

    def resblock(name, ci, co):
        if time_in:                                                            # DD:193-196
            S[f"{name}.mlp.1.weight"] = (co * 2, tdim)
            S[f"{name}.mlp.1.bias"] = (co * 2,)
        for blk, cin in (("block1", ci), ("block2", co)):
            conv(f"{name}.{blk}.proj", co, cin, 3)
            S[f"{name}.{blk}.norm.weight"] = (co,)
            S[f"{name}.{blk}.norm.bias"] = (co,)
        if ci != co:
            conv(f"{name}.res_conv", co, ci, 1)

    def linattn(name, c):
        S[f"{name}.fn.fn.to_qkv.weight"] = (HEADS * DIM_HEAD * 3, c, 1, 1)
        conv(f"{name}.fn.fn.to_out.0", c, HEADS * DIM_HEAD, 1)
        S[f"{name}.fn.fn.to_out.1.g"] = (1, c, 1, 1)
        S[f"{name}.fn.norm.g"] = (1, c, 1, 1)

    conv("init_conv", dim, channels, 7)
    if time_in:                                                                # DD:308-324
        S["time_mlp.1.weight"] = (tdim, dim)
        S["time_mlp.1.bias"] = (tdim,)
        S["time_mlp.3.weight"] = (tdim, tdim)
        S["time_mlp.3.bias"] = (tdim,)
    for i, (ci, co) in enumerate(in_out):
        resblock(f"downs.{i}.0", ci, ci)
        resblock(f"downs.{i}.1", ci, ci)
        linattn(f"downs.{i}.2", ci)
        if i < n - 1:
            conv(f"downs.{i}.3.1", co, ci * 4, 1)
        else:
            conv(f"downs.{i}.3", co, ci, 3)
    for i, (ci, co) in enumerate(reversed(in_out)):
        resblock(f"ups.{i}.0", co + ci, co)
        resblock(f"ups.{i}.1", co + ci, co)
        linattn(f"ups.{i}.2", co)
        if i < n - 1:
            conv(f"ups.{i}.3.1", ci, co, 3)
        else:
            conv(f"ups.{i}.3", ci, co, 3)
    # registration order of the reference: downs, ups (DD:328-329), then mid (DD:342-345)
    mid = dims[-1]
    resblock("mid_block1", mid, mid)
    S["mid_attn.fn.fn.to_qkv.weight"] = (HEADS * DIM_HEAD * 3, mid, 1, 1)
    conv("mid_attn.fn.fn.to_out", mid, HEADS * DIM_HEAD, 1)
    S["mid_attn.fn.norm.g"] = (1, mid, 1, 1)
    resblock("mid_block2", mid, mid)
    resblock("final_res_block", dim * 2, dim)
    conv("final_conv", out_dim, dim, 1)
    return S


def closed_form_params(shapes, amp=None):
    """Deterministic weight fill shared by the golden generator, the oracle and the HIP tests.

    w_k[i] = a_k * sin(0.37*i + k) (+1 for norm gains) with a_k scaled like a default init,
    so activations stay O(1) through the net and no weight file is needed.
    """
    P = {}
    for k, (name, shp) in enumerate(shapes.items()):
        numel = 1
        for s in shp:
            numel *= s
        i = torch.arange(numel, dtype=torch.float64)
        base = torch.sin(0.37 * i + k)
        if name.endswith(".g") or name.endswith("norm.weight"):
            v = 1.0 + 0.1 * base
        elif name.endswith("bias"):
            v = 0.05 * base
        else:
            fan_in = numel // shp[0]
            a = (amp if amp is not None else 1.0) * math.sqrt(3.0 / fan_in)
            v = a * base
        P[name] = v.to(torch.float32).reshape(shp)
    return P


def random_params(shapes, seed=0):
    """Well-conditioned weight fill: what ``torch.nn`` default initialisation looks like statistically (zero-mean uniform
    weights at 1/sqrt(fan_in) scale, small biases, gains near 1), drawn from numpy's counter-based Philox generator keyed
    on (seed, position in the state dict) so that the golden generator, the oracle and the HIP tests rebuild the very same
    35.7 M values without a weight file.  Under these weights the reference's own bf16-autocast run sits ~1e-2 from its
    fp32 run (the closed-form sin fill above amplifies bf16 rounding ten-fold)."""
    import numpy as np
    P = {}
    for k, (name, shp) in enumerate(shapes.items()):
        rng = np.random.Generator(np.random.Philox(key=[seed, k]))
        numel = 1
        for s in shp:
            numel *= s
        u = torch.from_numpy(rng.random(numel, dtype=np.float32) * 2.0 - 1.0)
        if name.endswith(".g") or name.endswith("norm.weight"):
            v = 1.0 + 0.1 * u
        elif name.endswith("bias"):
            v = 0.05 * u
        elif name.endswith("proj.weight"):
            # WeightStandardizedConv2d (DD:101-114) is invariant to the weight's scale except through eps: at unit
            # variance the reference's dtype-dependent eps (1e-5 vs 1e-3, DD:107) moves the output by 5e-4, so its fp32 and
            # bf16-autocast runs compute the same function and differ by bf16 rounding alone
            v = u * math.sqrt(3.0)
        elif name.endswith("to_out.0.weight"):
            # LinearAttention's output is O(1/n) (v / (h*w), DD:238), so at default scale the LayerNorm behind to_out.0
            # (DD:229-230) normalises a variance far below its eps and the dtype-dependent eps (DD:122) alone moves the UNet
            # output by 24 %.  Scaled up, that LayerNorm sees O(1)..O(40) inputs at the fixture sizes, eps is immaterial and
            # the attention path is numerically alive in the output.
            v = u * math.sqrt(3.0 / (numel // shp[0])) * (2.5e6 / shp[0])
        else:
            v = u * math.sqrt(3.0 / (numel // shp[0]))
        P[name] = v.to(torch.float32).reshape(shp)
    return P
