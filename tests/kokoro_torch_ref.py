"""An INDEPENDENT restatement of the published Kokoro-82M network (hexgrad/Kokoro-82M: kokoro/model.py, kokoro/modules.py, kokoro/istftnet.py; StyleTTS2 family) as plain
torch.nn modules — written from the published module list and forward passes, NOT from include/skw_kokoro_net.h, whose wiring it exists to check (the
product's HIP backend and its CPU checker instantiate one wiring template, so a swapped AdaIN input or a mis-recalled stride is invisible to a comparison of the two).

  * module and parameter names are the published ones, so the tensors tools/make_synth_kokoro.py writes load with `load_state_dict(strict=True)` — which also proves the
    product's name binding (weight-norm pairs folded into `.weight`, as an export folds them; `InstanceNorm1d(affine=False)` as StyleTTS2 defines AdaIN — Kokoro's affine=True
    is an ONNX-export workaround with identity parameters);
  * ALBERT is transformers' own AlbertModel (a second party's implementation);
  * every contraction is torch's (fp32, its own summation order), every statistic torch's, exp / tanh / sin libm's: agreement with the product is to ~1e-5, not bitwise;
  * what the published model draws from torch.rand / torch.randn is an INPUT here (`rand_ini`, `noise`), so that a caller can hand in the numbers the product's
    deterministic stand-in uses (tests/test_cpu_kokoro.py restates that counter hash in numpy); dropout is inference-mode identity.

Test infrastructure only (tests/golden/make_kokoro_torch_goldens.py writes the committed per-stage fixture from it; tests/test_cpu_kokoro.py, tests/test_gpu_kokoro.py)."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

STYLE_DIM, N_FFT, HOP, SAMPLE_RATE = 128, 20, 5, 24000
UPSAMPLE_RATES, UPSAMPLE_KERNELS = (10, 6), (20, 12)
RESBLOCK_KERNELS, RESBLOCK_DILATIONS = (3, 7, 11), ((1, 3, 5), (1, 3, 5), (1, 3, 5))
HARMONICS = 8


# ------------------------------------------------------------------ kokoro/modules.py
class LinearNorm(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim)

    def forward(self, x):
        return self.linear_layer(x)


class LayerNorm(nn.Module):
    """over the channel axis of [B, C, T]"""
    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels)); self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        return F.layer_norm(x.transpose(1, -1), (self.channels,), self.gamma, self.beta, self.eps).transpose(1, -1)


class TextEncoder(nn.Module):
    def __init__(self, channels, kernel_size, depth, n_symbols):
        super().__init__()
        self.embedding = nn.Embedding(n_symbols, channels)
        self.cnn = nn.ModuleList([nn.Sequential(nn.Conv1d(channels, channels, kernel_size, padding=(kernel_size - 1) // 2), LayerNorm(channels), nn.LeakyReLU(0.2), nn.Dropout(0.2))
                                  for _ in range(depth)])
        self.lstm = nn.LSTM(channels, channels // 2, 1, batch_first=True, bidirectional=True)

    def forward(self, ids):                     # [1, T] -> [1, C, T]   (one unpadded sequence: the masks of the published forward are all False)
        x = self.embedding(ids).transpose(1, 2)
        for c in self.cnn:
            x = c(x)
        x, _ = self.lstm(x.transpose(1, 2))
        return x.transpose(-1, -2)


class AdaLayerNorm(nn.Module):
    def __init__(self, style_dim, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.fc = nn.Linear(style_dim, channels * 2)

    def forward(self, x, s):                    # x [B, T, C]
        h = self.fc(s)
        gamma, beta = torch.chunk(h.view(h.size(0), h.size(1), 1), chunks=2, dim=1)
        gamma, beta = gamma.transpose(1, -1), beta.transpose(1, -1)
        x = F.layer_norm(x, (self.channels,), eps=self.eps)
        return (1 + gamma) * x + beta


class DurationEncoder(nn.Module):
    def __init__(self, sty_dim, d_model, nlayers):
        super().__init__()
        self.lstms = nn.ModuleList()
        for _ in range(nlayers):
            self.lstms.append(nn.LSTM(d_model + sty_dim, d_model // 2, num_layers=1, batch_first=True, bidirectional=True))
            self.lstms.append(AdaLayerNorm(sty_dim, d_model))

    def forward(self, x, style):                # x [1, d, T] -> [1, T, d + sty]
        T = x.shape[-1]
        s = style.unsqueeze(1).expand(-1, T, -1)                 # [1, T, sty]
        x = torch.cat([x.transpose(1, 2), s], dim=-1)            # [1, T, d + sty]
        for block in self.lstms:
            if isinstance(block, AdaLayerNorm):
                x = torch.cat([block(x, style), s], dim=-1)
            else:
                x, _ = block(x)
        return x


class ProsodyPredictor(nn.Module):
    def __init__(self, style_dim, d_hid, nlayers, max_dur):
        super().__init__()
        self.text_encoder = DurationEncoder(style_dim, d_hid, nlayers)
        self.lstm = nn.LSTM(d_hid + style_dim, d_hid // 2, 1, batch_first=True, bidirectional=True)
        self.duration_proj = LinearNorm(d_hid, max_dur)
        self.shared = nn.LSTM(d_hid + style_dim, d_hid // 2, 1, batch_first=True, bidirectional=True)
        self.F0 = nn.ModuleList([AdainResBlk1d(d_hid, d_hid, style_dim), AdainResBlk1d(d_hid, d_hid // 2, style_dim, upsample=True), AdainResBlk1d(d_hid // 2, d_hid // 2, style_dim)])
        self.N = nn.ModuleList([AdainResBlk1d(d_hid, d_hid, style_dim), AdainResBlk1d(d_hid, d_hid // 2, style_dim, upsample=True), AdainResBlk1d(d_hid // 2, d_hid // 2, style_dim)])
        self.F0_proj = nn.Conv1d(d_hid // 2, 1, 1, 1, 0)
        self.N_proj = nn.Conv1d(d_hid // 2, 1, 1, 1, 0)

    def F0Ntrain(self, x, s):                   # x [1, d + sty, F]
        x, _ = self.shared(x.transpose(-1, -2))
        F0 = x.transpose(-1, -2)
        for block in self.F0:
            F0 = block(F0, s)
        N = x.transpose(-1, -2)
        for block in self.N:
            N = block(N, s)
        return self.F0_proj(F0).squeeze(1), self.N_proj(N).squeeze(1)


# ------------------------------------------------------------------ kokoro/istftnet.py
class AdaIN1d(nn.Module):
    def __init__(self, style_dim, num_features):
        super().__init__()
        self.norm = nn.InstanceNorm1d(num_features, affine=False)
        self.fc = nn.Linear(style_dim, num_features * 2)

    def forward(self, x, s):
        h = self.fc(s)
        gamma, beta = torch.chunk(h.view(h.size(0), h.size(1), 1), chunks=2, dim=1)
        return (1 + gamma) * self.norm(x) + beta


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class AdaINResBlock1(nn.Module):
    def __init__(self, channels, kernel_size, dilation, style_dim):
        super().__init__()
        self.convs1 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d)) for d in dilation])
        self.convs2 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=1, padding=get_padding(kernel_size, 1)) for _ in dilation])
        self.adain1 = nn.ModuleList([AdaIN1d(style_dim, channels) for _ in dilation])
        self.adain2 = nn.ModuleList([AdaIN1d(style_dim, channels) for _ in dilation])
        self.alpha1 = nn.ParameterList([nn.Parameter(torch.ones(channels)) for _ in dilation])      # (published shape [1, C, 1]; the files carry [C])
        self.alpha2 = nn.ParameterList([nn.Parameter(torch.ones(channels)) for _ in dilation])

    def forward(self, x, s):
        for c1, c2, n1, n2, a1, a2 in zip(self.convs1, self.convs2, self.adain1, self.adain2, self.alpha1, self.alpha2):
            a1, a2 = a1.view(1, -1, 1), a2.view(1, -1, 1)
            xt = n1(x, s)
            xt = xt + (1 / a1) * (torch.sin(a1 * xt) ** 2)       # Snake1D
            xt = c1(xt)
            xt = n2(xt, s)
            xt = xt + (1 / a2) * (torch.sin(a2 * xt) ** 2)
            xt = c2(xt)
            x = xt + x
        return x


class SineGen(nn.Module):
    def __init__(self, samp_rate, upsample_scale, harmonic_num, sine_amp=0.1, noise_std=0.003, voiced_threshold=0):
        super().__init__()
        self.sine_amp, self.noise_std, self.harmonic_num, self.sampling_rate, self.voiced_threshold, self.upsample_scale = sine_amp, noise_std, harmonic_num, samp_rate, voiced_threshold, upsample_scale

    def _f02sine(self, f0_values, rand_ini):     # [1, L, dim]
        # (the published statements, evaluated in float64: in float32 a phase of 1e5 rad has an ulp of 0.008 rad and torch's own result is only that good)
        f0_values, rand_ini = f0_values.double(), rand_ini.double()
        rad_values = (f0_values / self.sampling_rate) % 1
        rad_values = rad_values.clone()
        rad_values[:, 0, :] = rad_values[:, 0, :] + rand_ini
        rad_values = F.interpolate(rad_values.transpose(1, 2), scale_factor=1 / self.upsample_scale, mode="linear").transpose(1, 2)
        phase = torch.cumsum(rad_values, dim=1) * 2 * torch.pi
        phase = F.interpolate(phase.transpose(1, 2) * self.upsample_scale, scale_factor=self.upsample_scale, mode="linear").transpose(1, 2)
        return torch.sin(phase).float()

    def forward(self, f0, rand_ini, noise):      # f0 [1, L, 1]; rand_ini [1, dim] (element 0 is zero in the published code); noise [1, L, dim] standing in for randn_like
        fn = f0.double() * torch.arange(1, self.harmonic_num + 2, dtype=torch.float64).view(1, 1, -1)
        sine_waves = self._f02sine(fn, rand_ini) * self.sine_amp
        uv = (f0 > self.voiced_threshold).to(f0.dtype)
        noise_amp = uv * self.noise_std + (1 - uv) * self.sine_amp / 3
        return sine_waves * uv + noise_amp * noise, uv


class SourceModuleHnNSF(nn.Module):
    def __init__(self, sampling_rate, upsample_scale, harmonic_num, voiced_threshod):
        super().__init__()
        self.l_sin_gen = SineGen(sampling_rate, upsample_scale, harmonic_num, 0.1, 0.003, voiced_threshod)
        self.l_linear = nn.Linear(harmonic_num + 1, 1)

    def forward(self, x, rand_ini, noise):
        sine_wavs, uv = self.l_sin_gen(x, rand_ini, noise)
        return torch.tanh(self.l_linear(sine_wavs)), uv


class Generator(nn.Module):
    def __init__(self, style_dim, upsample_initial_channel):
        super().__init__()
        self.num_kernels, self.num_upsamples = len(RESBLOCK_KERNELS), len(UPSAMPLE_RATES)
        scale = math.prod(UPSAMPLE_RATES) * HOP
        self.m_source = SourceModuleHnNSF(SAMPLE_RATE, scale, HARMONICS, 10)
        self.f0_upsamp = nn.Upsample(scale_factor=scale)
        self.noise_convs, self.noise_res, self.ups, self.resblocks = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i, (u, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNELS)):
            self.ups.append(nn.ConvTranspose1d(upsample_initial_channel // (2 ** i), upsample_initial_channel // (2 ** (i + 1)), k, u, padding=(k - u) // 2))
        for i in range(len(self.ups)):
            ch = upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(RESBLOCK_KERNELS, RESBLOCK_DILATIONS):
                self.resblocks.append(AdaINResBlock1(ch, k, d, style_dim))
            if i + 1 < len(UPSAMPLE_RATES):
                stride_f0 = math.prod(UPSAMPLE_RATES[i + 1:])
                self.noise_convs.append(nn.Conv1d(N_FFT + 2, ch, kernel_size=stride_f0 * 2, stride=stride_f0, padding=(stride_f0 + 1) // 2))
                self.noise_res.append(AdaINResBlock1(ch, 7, (1, 3, 5), style_dim))
            else:
                self.noise_convs.append(nn.Conv1d(N_FFT + 2, ch, kernel_size=1))
                self.noise_res.append(AdaINResBlock1(ch, 11, (1, 3, 5), style_dim))
        self.conv_post = nn.Conv1d(ch, N_FFT + 2, 7, 1, padding=3)
        self.reflection_pad = nn.ReflectionPad1d((1, 0))
        self.register_buffer("window", torch.hann_window(N_FFT, periodic=True), persistent=False)

    def forward(self, x, s, f0, rand_ini, noise, taps, har_given=None):
        f0 = self.f0_upsamp(f0[:, None]).transpose(1, 2)         # [1, L, 1]
        har_source, _ = self.m_source(f0, rand_ini, noise)
        har_source = har_source.transpose(1, 2).squeeze(1)
        st = torch.stft(har_source, N_FFT, HOP, N_FFT, window=self.window, return_complex=True)
        har = torch.cat([torch.abs(st), torch.angle(st)], dim=1)
        taps["har"] = har[0].transpose(0, 1)
        if har_given is not None:                                # teacher forcing at the source spectrum (its phase channel can wrap at +-pi on an ulp: see forward_with_tokens)
            har = torch.as_tensor(har_given, dtype=torch.float32).transpose(0, 1).unsqueeze(0)
        for i in range(self.num_upsamples):
            x = F.leaky_relu(x, negative_slope=0.1)
            x_source = self.noise_res[i](self.noise_convs[i](har), s)
            x = self.ups[i](x)
            if i == self.num_upsamples - 1:
                x = self.reflection_pad(x)
            x = x + x_source
            xs = None
            for j in range(self.num_kernels):
                r = self.resblocks[i * self.num_kernels + j](x, s)
                xs = r if xs is None else xs + r
            x = xs / self.num_kernels
        x = F.leaky_relu(x)
        x = self.conv_post(x)
        taps["post"] = x[0].transpose(0, 1)
        spec = torch.exp(x[:, :N_FFT // 2 + 1, :])
        phase = torch.sin(x[:, N_FFT // 2 + 1:, :])
        return torch.istft(spec * torch.exp(phase * 1j), N_FFT, HOP, N_FFT, window=self.window)


class UpSample1d(nn.Module):
    def __init__(self, layer_type):
        super().__init__()
        self.layer_type = layer_type

    def forward(self, x):
        return x if self.layer_type == "none" else F.interpolate(x, scale_factor=2, mode="nearest")


class AdainResBlk1d(nn.Module):
    def __init__(self, dim_in, dim_out, style_dim, upsample=False):
        super().__init__()
        self.upsample_type = "upsample" if upsample else "none"
        self.upsample = UpSample1d(self.upsample_type)
        self.learned_sc = dim_in != dim_out
        self.conv1 = nn.Conv1d(dim_in, dim_out, 3, 1, 1)
        self.conv2 = nn.Conv1d(dim_out, dim_out, 3, 1, 1)
        self.norm1 = AdaIN1d(style_dim, dim_in)
        self.norm2 = AdaIN1d(style_dim, dim_out)
        if self.learned_sc:
            self.conv1x1 = nn.Conv1d(dim_in, dim_out, 1, 1, 0, bias=False)
        self.pool = nn.ConvTranspose1d(dim_in, dim_in, kernel_size=3, stride=2, groups=dim_in, padding=1, output_padding=1) if upsample else nn.Identity()

    def _shortcut(self, x):
        x = self.upsample(x)
        return self.conv1x1(x) if self.learned_sc else x

    def _residual(self, x, s):
        x = F.leaky_relu(self.norm1(x, s), 0.2)
        x = self.pool(x)
        x = self.conv1(x)
        x = F.leaky_relu(self.norm2(x, s), 0.2)
        return self.conv2(x)

    def forward(self, x, s):
        return (self._residual(x, s) + self._shortcut(x)) * torch.rsqrt(torch.tensor(2.0))


class Decoder(nn.Module):
    def __init__(self, dim_in, style_dim, dec_c, asr_c, gen_c0, n_decode):
        super().__init__()
        self.encode = AdainResBlk1d(dim_in + 2, dec_c, style_dim)
        self.decode = nn.ModuleList([AdainResBlk1d(dec_c + 2 + asr_c, gen_c0 if i + 1 == n_decode else dec_c, style_dim, upsample=(i + 1 == n_decode)) for i in range(n_decode)])
        self.F0_conv = nn.Conv1d(1, 1, kernel_size=3, stride=2, groups=1, padding=1)
        self.N_conv = nn.Conv1d(1, 1, kernel_size=3, stride=2, groups=1, padding=1)
        self.asr_res = nn.Sequential(nn.Conv1d(dim_in, asr_c, kernel_size=1))
        self.generator = Generator(style_dim, gen_c0)

    def forward(self, asr, F0_curve, N, s, rand_ini, noise, taps, har_given=None):
        F0 = self.F0_conv(F0_curve.unsqueeze(1))
        Nc = self.N_conv(N.unsqueeze(1))
        x = self.encode(torch.cat([asr, F0, Nc], dim=1), s)
        asr_res = self.asr_res(asr)
        res = True
        for block in self.decode:
            if res:
                x = torch.cat([x, asr_res, F0, Nc], dim=1)
            x = block(x, s)
            if block.upsample_type != "none":
                res = False
        taps["dec"] = x[0].transpose(0, 1)
        return self.generator(x, s, F0_curve, rand_ini, noise, taps, har_given)


# ------------------------------------------------------------------ kokoro/model.py
class KModel(nn.Module):
    def __init__(self, n_token, emb, hid, ffn, layers, heads, max_pos, d, max_dur, te_depth, te_kernel, dec_c, asr_c, gen_c0, n_decode):
        super().__init__()
        from transformers import AlbertConfig, AlbertModel
        self.bert = AlbertModel(AlbertConfig(vocab_size=n_token, embedding_size=emb, hidden_size=hid, num_attention_heads=heads, intermediate_size=ffn, num_hidden_layers=layers,
                                             max_position_embeddings=max_pos, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), add_pooling_layer=False)
        self.bert_encoder = nn.Linear(hid, d)
        self.predictor = ProsodyPredictor(STYLE_DIM, d, 3, max_dur)
        self.text_encoder = TextEncoder(d, te_kernel, te_depth, n_token)
        self.decoder = Decoder(d, STYLE_DIM, dec_c, asr_c, gen_c0, n_decode)

    @torch.no_grad()
    def forward_with_tokens(self, input_ids, ref_s, speed=1.0, rand_ini=None, noise_fn=None, curves=None, har=None):
        """input_ids [1, T] (pad id 0 at both ends), ref_s [1, 256] -> (audio [600 F], pred_dur [T], per-stage taps).
        curves = (F0 [2 F], N [2 F]): teacher forcing at the curve stage — the decoder and the harmonic source take THESE instead of the predictor's own (the taps still show the
        predictor's).  The source integrates F0 into a phase, so two implementations whose F0 curves differ by 1e-6 differ by 1e-2 in everything after it; with the curves
        handed over, the decoder and generator are compared on equal inputs.
        har = the source spectrum [rows, 22] (11 magnitudes + 11 phases): teacher forcing one stage later — the generator body takes THIS (the tap still shows the model's own).
        A phase that sits at +-pi comes out on either side on an ulp (the first, reflect-padded frame is symmetric: its imaginary parts are rounding noise), which is a 2 pi step
        in a network input, and the instance norms spread its effect over the whole utterance."""
        taps = {}
        bert_dur = self.bert(input_ids, attention_mask=torch.ones_like(input_ids)).last_hidden_state
        taps["bert"] = bert_dur[0]
        d_en = self.bert_encoder(bert_dur).transpose(-1, -2)
        taps["d_en"] = d_en[0].transpose(0, 1)
        s = ref_s[:, 128:]
        d = self.predictor.text_encoder(d_en, s)
        x, _ = self.predictor.lstm(d)
        duration = torch.sigmoid(self.predictor.duration_proj(x)).sum(dim=-1) / speed
        pred_dur = torch.round(duration).clamp(min=1).long().squeeze(0)
        indices = torch.repeat_interleave(torch.arange(input_ids.shape[1]), pred_dur)
        aln = torch.zeros((input_ids.shape[1], indices.shape[0]))
        aln[indices, torch.arange(indices.shape[0])] = 1
        aln = aln.unsqueeze(0)
        en = d.transpose(-1, -2) @ aln
        F0_pred, N_pred = self.predictor.F0Ntrain(en, s)
        taps["f0"], taps["en"] = F0_pred[0], N_pred[0]
        if curves is not None:
            F0_pred, N_pred = torch.as_tensor(curves[0], dtype=torch.float32).view(1, -1), torch.as_tensor(curves[1], dtype=torch.float32).view(1, -1)
        t_en = self.text_encoder(input_ids)
        taps["t_en"] = t_en[0].transpose(0, 1)
        asr = t_en @ aln
        L = F0_pred.shape[-1] * math.prod(UPSAMPLE_RATES) * HOP
        ri = torch.zeros(1, HARMONICS + 1) if rand_ini is None else rand_ini
        noise = torch.zeros(1, L, HARMONICS + 1) if noise_fn is None else noise_fn(L)
        audio = self.decoder(asr, F0_pred, N_pred, ref_s[:, :128], ri, noise, taps, har).squeeze()
        return audio, pred_dur, taps


def build_from_tensors(w):
    """w: {published module name: np.ndarray} as tools/make_synth_kokoro.py writes them (tests/onnx_mini.py reads model.onnx).  Geometry is read from the tensors' shapes the way a
    config.json would state it; returns the model with every tensor loaded strict=True."""
    w = dict(w)
    layers = int(round(float(np.asarray(w.pop("bert.config.num_hidden_layers")).ravel()[0])))      # (the synthetic files carry this one config value as a tensor)
    emb = w["bert.embeddings.word_embeddings.weight"].shape[1]
    hid, ffn = w["bert.encoder.embedding_hidden_mapping_in.weight"].shape[0], w["bert.encoder.albert_layer_groups.0.albert_layers.0.ffn.weight"].shape[0]
    d = w["bert_encoder.weight"].shape[0]
    n_decode = 1 + max(int(k.split(".")[2]) for k in w if k.startswith("decoder.decode."))
    te_depth = 1 + max(int(k.split(".")[2]) for k in w if k.startswith("text_encoder.cnn."))
    m = KModel(n_token=w["bert.embeddings.word_embeddings.weight"].shape[0], emb=emb, hid=hid, ffn=ffn, layers=layers, heads=max(1, hid // 64),
               max_pos=w["bert.embeddings.position_embeddings.weight"].shape[0], d=d, max_dur=w["predictor.duration_proj.linear_layer.weight"].shape[0], te_depth=te_depth,
               te_kernel=w["text_encoder.cnn.0.0.weight"].shape[2], dec_c=w["decoder.encode.conv1.weight"].shape[0], asr_c=w["decoder.asr_res.0.weight"].shape[0],
               gen_c0=w["decoder.generator.ups.0.weight"].shape[0], n_decode=n_decode).eval().float()
    sd = {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in w.items()}
    m.load_state_dict(sd, strict=True)
    return m
