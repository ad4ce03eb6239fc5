"""hcir.pipeline.StreamPipeline (two query batches in flight on two HIP streams, one engine buffer slot each): the
results are those of the batches run one after the other — embeddings bit for bit, top-k values and indices equal."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_stream_pipeline_equals_sequential(hcir_built):
    from hcir import vit_engine
    from hcir.gallery import ResidentGallery
    from hcir.main_backbone import SHAM2
    from hcir.pipeline import StreamPipeline
    keep = vit_engine.DEFAULT_RESID_DTYPE
    vit_engine.DEFAULT_RESID_DTYPE = torch.float16
    try:
        torch.manual_seed(3)
        model = SHAM2("vit_b_16").eval().cuda()
        g = F.normalize(torch.randn(20_000, 768, device="cuda"), dim=1)
        gal = ResidentGallery(g)
        batches = [torch.randn(n, 3, 224, 224, device="cuda") for n in (64, 64, 64, 64, 64, 64, 64)]
        ref = []
        with torch.no_grad():
            for xb in batches:
                e32, e16 = model.backbone.forward_cls(xb, l2_normalize=True, want_f16=True)
                ref.append((e32.clone(),) + tuple(t.clone() for t in gal.search_begin(e32, 10, q16=e16).finish()))
        pipe = StreamPipeline(model.backbone, gal, 10, depth=2)
        got = []
        for xb in batches:
            r = pipe.submit(xb)
            if r is not None:
                got.append(r)
        got += pipe.drain()
        assert len(got) == len(batches)
        for (e, v, i), (gv, gi) in zip(ref, got):
            assert torch.equal(v, gv) and torch.equal(i, gi)
        # a second pass over the same pipeline object (slots re-used), against the oracle's definition of the top-k
        r = [pipe.submit(batches[0]), pipe.submit(batches[1])] + pipe.drain()
        r = [t for t in r if t is not None]
        assert torch.equal(r[-2][1], ref[0][2]) and torch.equal(r[-1][1], ref[1][2])
        s = ref[0][0].double() @ g.double().t()
        assert torch.equal(torch.sort(s, 1, descending=True)[1][:, :10], ref[0][2])
    finally:
        vit_engine.DEFAULT_RESID_DTYPE = keep
