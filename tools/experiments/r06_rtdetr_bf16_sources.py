"""Round 6: where does the bf16 error of config 5 (yolov3-rtdetr, bs 16) come from?  The decoder runs on the ORACLE'S top-300 queries
(`RTDETRDecoder.query_override`) so rows compare one to one; variants switch parts of the perf mode back to exact float32:
  perf      bf16 backbone, bf16 input projections, bf16-product linears, bf16 value rows          (the quoted mode)
  lin32     ... with every nn.Linear of the head in exact float32 (linear_bf16 = False)
  dec32     ... and float32 input projections / value projections too: ONLY the backbone is bf16
  f32       everything float32 (the parity mode) - the floor of the comparison
Prints per variant: encoder-side per-token class-probability / box deviations, decoder row deviations.  GPU only."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import tests.test_hip_e2e as T  # noqa: E402
from ultralytics_pro_amd.nn.modules import rtdetr as RT  # noqa: E402
from ultralytics_pro_amd.utils import procedural as P  # noqa: E402

DEV = torch.device("cuda:0")


def run(family, variant, bs=16):
    x, y_ref, ref = T._oracle_rtdetr_taps(bs, family)
    dt = torch.float32 if variant == "f32" else torch.bfloat16
    m = T._build("yolov3-rtdetr", dt, family=family)
    head = m.model[-1]
    if variant in ("lin32", "dec32"):
        head.linear_bf16 = False
    if variant == "dec32":
        head.proj_bf16 = False
        head.fuse_value_proj = False
    head.taps = {}
    head.query_override = ref["topk"]
    with torch.no_grad(), T._dispatch("throughput"):
        y = m(x.to(DEV).to(dt).contiguous())[0]
        t = head.taps
        st, b = t["static"], t["bs"]
        sc = head.level_major_to_image(t["enc_scores"], st, b).float().cpu()
        saved = RT._LINEAR_BF16[0]
        RT._LINEAR_BF16[0] = bool(head.linear_bf16 and dt == torch.bfloat16)
        try:
            delta = head.enc_bbox_head(t["features"], key="all_tokens")
        finally:
            RT._LINEAR_BF16[0] = saved
        delta = head.level_major_to_image(delta, st, b).float().cpu()
        feat = head.level_major_to_image(t["features"], st, b).float().cpu()
    torch.cuda.synchronize()
    valid = ref["valid"]
    box = (delta + st["anchors"].cpu().view(1, -1, 4)).sigmoid()
    dp = (sc.sigmoid() - ref["scores"].sigmoid()).abs()
    db = (box - ref["enc_box"]).abs()[:, valid] * 640
    df = (feat - ref["features"]).abs()
    yc = y.float().cpu()
    rb = (yc[..., :4] - y_ref[..., :4]).abs().amax(2) * 640
    rs = (yc[..., 4:] - y_ref[..., 4:]).abs().amax(2)
    q = lambda v, p: float(v.flatten()[::3].quantile(p))
    print(f"{str(family):22s} {variant:6s} | enc: feature |d| max {df.max():.4f} (|f| max {ref['features'].abs().max():.2f}); prob |d| max {dp.max():.5f}; "
          f"box p50 {q(db, .5):.3f} p99 {q(db, .99):.3f} max {db.max():.3f} px, inside 0.5 px {float((db.amax(2) <= 0.5).float().mean()):.4f} "
          f"| dec rows: box p50 {q(rb, .5):.3f} p99 {q(rb, .99):.3f} max {rb.max():.3f} px, score p99 {q(rs, .99):.5f} max {rs.max():.5f}, "
          f"inside 0.5/0.01 {float(((rb <= 0.5) & (rs <= 0.01)).float().mean()):.4f}, inside 4/0.01 {float(((rb <= 4) & (rs <= 0.01)).float().mean()):.4f}",
          flush=True)


if __name__ == "__main__":
    for fam in ("smooth:yolov3-rtdetr", None):
        for v in ("perf", "lin32", "dec32", "f32"):
            run(fam, v)
