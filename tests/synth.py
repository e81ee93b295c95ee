"""Deterministic synthetic inputs shared by tests/ and bench.py.

No model weights or tokenizer files exist offline, so every table, token stream
and corpus is generated from fixed seeds.
"""
import numpy as np

DIM = 256


def unit_rows(n, seed, dup_frac=0.01, zero_frac=0.001):
    """n unit-normalised N(0,1) rows, with exact-duplicate rows (ties) and all-zero rows
    (empty-line embeddings) planted at seeded positions.  Config c2's corpus generator."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, DIM), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    n_dup = int(n * dup_frac)
    if n_dup and n > 1:
        dst = rng.choice(n, size=n_dup, replace=False)
        src = rng.integers(0, n, size=n_dup)
        x[dst] = x[src]
    n_zero = int(n * zero_frac)
    if n_zero:
        x[rng.choice(n, size=n_zero, replace=False)] = 0.0
    return np.ascontiguousarray(x, dtype=np.float32)


def unit_query(seed, nq=1):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((nq, DIM), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    return np.ascontiguousarray(q, dtype=np.float32)


def table(V, seed=2, scale=0.1):
    """Synthetic stand-in for the potion-multilingual-128M `embeddings` tensor [V x 256]."""
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(rng.standard_normal((V, DIM), dtype=np.float32) * np.float32(scale))


def token_lines(n_lines, V, seed=1, min_tok=0, max_tok=24, zipf_s=1.1):
    """CSR token stream: Zipf-distributed ids (hot rows, like real text), ragged lengths
    including empty lines.  Returns (ids uint32[T], offsets uint64[n_lines+1])."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_tok, max_tok + 1, size=n_lines)
    T = int(lens.sum())
    ranks = rng.zipf(zipf_s, size=T).astype(np.int64)
    ids = ((ranks - 1) % V).astype(np.uint32)
    offsets = np.zeros(n_lines + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    return ids, offsets


def pseudo_prose(n_lines, vocab_size=50000, seed=1, min_words=5, max_words=20):
    """Config c1's 'plaintext lines': words w<id> drawn Zipf(1.1) from a synthetic vocab."""
    rng = np.random.default_rng(seed)
    lines = []
    for _ in range(n_lines):
        k = int(rng.integers(min_words, max_words + 1))
        ranks = (rng.zipf(1.1, size=k) - 1) % vocab_size
        lines.append(" ".join(f"w{int(r)}" for r in ranks))
    return lines


def clustered_rows_torch(n, n_centers, latent, seed, device, spread=0.35, noise=0.01):
    """Config c5's clustered corpus, generated on the device (torch): row = topic centre + spread * z . B_topic
    + noise, normalised; z ~ N(0, I_latent), B_topic a per-topic latent x 256 basis.  Nearest neighbours of a row
    are the rows of its topic that are close in the latent space -- a graded, well-defined top-k with structure a
    quantiser can encode (isotropic 256-d noise has none)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    centers = torch.randn(n_centers, 256, device=device, generator=g)
    centers /= centers.norm(dim=1, keepdim=True)
    basis = torch.randn(n_centers, latent, 256, device=device, generator=g) / 16.0
    which = torch.randint(0, n_centers, (n,), device=device, generator=g)
    x = torch.empty(n, 256, device=device)
    step = 500_000
    for b in range(0, n, step):
        e = min(n, b + step)
        z = torch.randn(e - b, 1, latent, device=device, generator=g) * (spread / latent ** 0.5)
        xs = centers[which[b:e]] + torch.bmm(z, basis[which[b:e]]).squeeze(1)
        xs += (noise / 16.0) * torch.randn(e - b, 256, device=device, generator=g)
        x[b:e] = xs / xs.norm(dim=1, keepdim=True)
    return x


def clustered_model_torch(n_centers, latent, seed, device):
    """The generative model behind clustered_rows_torch, kept so that QUERIES can be drawn from it independently of
    the corpus rows (fresh latent draws of random topics -- not perturbed corpus rows)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    centers = torch.randn(n_centers, 256, device=device, generator=g)
    centers /= centers.norm(dim=1, keepdim=True)
    basis = torch.randn(n_centers, latent, 256, device=device, generator=g) / 16.0
    return dict(centers=centers, basis=basis, latent=latent, device=device)


def clustered_sample_torch(model, n, seed, spread=0.35, noise=0.01):
    import torch

    g = torch.Generator(device=model["device"])
    g.manual_seed(seed)
    centers, basis, latent = model["centers"], model["basis"], model["latent"]
    which = torch.randint(0, centers.shape[0], (n,), device=model["device"], generator=g)
    x = torch.empty(n, 256, device=model["device"])
    step = 500_000
    for b in range(0, n, step):
        e = min(n, b + step)
        z = torch.randn(e - b, 1, latent, device=model["device"], generator=g) * (spread / latent ** 0.5)
        xs = centers[which[b:e]] + torch.bmm(z, basis[which[b:e]]).squeeze(1)
        xs += (noise / 16.0) * torch.randn(e - b, 256, device=model["device"], generator=g)
        x[b:e] = xs / xs.norm(dim=1, keepdim=True)
    return x
