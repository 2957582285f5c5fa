"""tools/eval_on_tao_amodal.py: HeldLogs -- what the worker thread logs while
the image level runs is written after it, on the handlers it was bound for."""
import importlib.util
import io
import logging
import os
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cli():
    spec = importlib.util.spec_from_file_location(
        "cli_heldlogs", os.path.join(ROOT, "tools", "eval_on_tao_amodal.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _loggers():
    main = logging.getLogger("heldlogs.main")
    other = logging.getLogger("heldlogs.other")
    out_main, out_root = io.StringIO(), io.StringIO()
    h_main, h_root = logging.StreamHandler(out_main), logging.StreamHandler(out_root)
    for lg in (main, other):
        lg.setLevel(logging.INFO)
        lg.propagate = True
    main.addHandler(h_main)
    logging.getLogger().addHandler(h_root)
    return main, other, out_main, out_root, h_main, h_root


def test_worker_records_follow_the_main_threads(monkeypatch):
    cli = _cli()
    main, other, out_main, out_root, h_main, h_root = _loggers()
    try:
        held = cli.HeldLogs(main)
        started, go = threading.Event(), threading.Event()

        def work():
            main.info("track 1")
            other.warning("track warning")
            started.set()
            go.wait(5)
            main.info("track 2")
            return 7

        box = {}
        t = threading.Thread(target=lambda: box.setdefault("r", held.run(work)))
        t.start()
        assert started.wait(5)
        main.info("image 1")          # the main thread is not held
        other.warning("image warning")
        go.set()
        t.join(5)
        assert box["r"] == 7
        assert out_main.getvalue().splitlines() == ["image 1"]
        held.close(replay=True)
        assert out_main.getvalue().splitlines() == ["image 1", "track 1", "track 2"]
        # the root handler saw every record of both loggers, the worker's last
        assert out_root.getvalue().splitlines() == [
            "image 1", "image warning", "track 1", "track warning", "track 2"]
        main.info("after")            # the gates are gone
        assert out_main.getvalue().splitlines()[-1] == "after"
    finally:
        main.removeHandler(h_main)
        logging.getLogger().removeHandler(h_root)


def test_dropped_when_the_image_level_failed():
    cli = _cli()
    main, other, out_main, out_root, h_main, h_root = _loggers()
    try:
        held = cli.HeldLogs(main)
        t = threading.Thread(target=lambda: held.run(lambda: main.info("track")))
        t.start()
        t.join(5)
        held.close(replay=False)
        assert out_main.getvalue() == "" and out_root.getvalue() == ""
        assert not any(isinstance(f, cli.HeldLogs._Gate) for f in h_main.filters)
    finally:
        main.removeHandler(h_main)
        logging.getLogger().removeHandler(h_root)
