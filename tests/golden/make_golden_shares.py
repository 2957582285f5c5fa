"""Golden text of the reference CLI on a prediction file that leaves one
rank's block of a two-rank run WITHOUT predictions (such a rank
used to die in the table build and the others hung in the exchange).

The by-video partition gives rank r the r-th block of the sorted image ids
(image level) and of the sorted video ids (track level).  ``lower_half`` keeps
the predictions whose image AND video lie in the first block of two, so rank 1
holds ground truth but no prediction at either level.  Writes
tests/golden/<name>/lower_half/{pred.json,cli_stdout.txt,cli_log.txt}.

Development container only (needs /root/reference)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402

CLI = os.path.join(refenv.REF, "tools", "eval_on_tao_amodal.py")


def lower_half(gt, preds):
    imgs = sorted(i["id"] for i in gt["images"])
    vids = sorted(v["id"] for v in gt["videos"])
    # (block_owner of evaluation/_dist.py: block 0 of 2 = positions [0, n // 2))
    keep_i = set(imgs[: len(imgs) // 2])
    keep_v = set(vids[: len(vids) // 2])
    return [p for p in preds if p["image_id"] in keep_i and p["video_id"] in keep_v]


def run(name):
    src = os.path.join(HERE, name)
    out = os.path.join(src, "lower_half")
    os.makedirs(out, exist_ok=True)
    gt_path = os.path.join(src, "gt.json")
    gt = json.load(open(gt_path))
    preds = lower_half(gt, json.load(open(os.path.join(src, "pred.json"))))
    assert preds
    pred_path = os.path.join(out, "pred.json")
    with open(pred_path, "w") as f:
        json.dump(preds, f, separators=(",", ":"))
    log = os.path.join(out, "cli_log.txt")
    r = subprocess.run(
        [sys.executable, CLI, "--track_result", pred_path, "--annotation", gt_path,
         "--output_log", log], cwd=os.path.dirname(CLI), env=refenv.cli_env(),
        capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(os.path.join(out, "cli_stdout.txt"), "w") as f:
        f.write(r.stdout)
    txt = open(log).read().replace(out + os.sep, "<PRED>/").replace(src + os.sep, "<DIR>/")
    with open(log, "w") as f:
        f.write(txt)
    print(name, len(preds), "predictions kept;", r.stdout.splitlines()[0])


if __name__ == "__main__":
    for n in sys.argv[1:] or ["f1", "f5"]:
        run(n)
