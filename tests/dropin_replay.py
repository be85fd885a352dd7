"""Replays tests/golden/script_traces.json (the calls the reference's own, unmodified training
scripts make on `hugectr`, recorded by tests/golden/make_script_traces.py) against a module."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def load_traces():
    with open(os.path.join(HERE, "golden", "script_traces.json")) as f:
        return json.load(f)


def _resolve(mod, path):
    o = mod
    for part in path.split("."):
        o = getattr(o, part)
    return o


class ReplayCallback:
    """mixin of the stand-in for a script-side hugectr.TrainingCallback subclass (its code cannot
    travel as data; its class name and plain attributes do): counts what fit() tells it"""

    def __init__(self, name, attrs):
        self.name, self.attrs = name, dict(attrs)
        self.events = []

    def on_training_start(self):
        self.events.append(("training_start",))

    def on_training_end(self, current_iter):
        self.events.append(("training_end", current_iter))

    def on_eval_start(self, current_iter):
        self.events.append(("eval_start", current_iter))
        return False

    def on_eval_end(self, current_iter, eval_results):
        self.events.append(("eval_end", current_iter, dict(eval_results)))
        return False


def make_callback(mod, v):
    cls = type(v["callback"], (ReplayCallback, mod.TrainingCallback), {})
    return cls(v["callback"], v["attrs"])


def replay(mod, calls, before=None, substitute=None):
    """executes the recorded calls on `mod`; before(i, target, args, kwargs) may edit args / kwargs
    in place (test-speed caps); substitute(target) may return a callable that takes the place of
    a module-level name (the CPU test's Model stand-in).  Returns the list of results."""
    results = []

    def dec(v):
        if isinstance(v, list):
            return [dec(x) for x in v]
        if isinstance(v, dict):
            if "name" in v and len(v) == 1:
                return _resolve(mod, v["name"])
            if "ref" in v and len(v) == 1:
                return results[v["ref"]]
            if "dict" in v and len(v) == 1:
                return {k: dec(x) for k, x in v["dict"].items()}
            if "pairs" in v and len(v) == 1:
                return {dec(k): dec(x) for k, x in v["pairs"]}
            if "callback" in v:
                return make_callback(mod, v)
            raise ValueError(v)
        return v

    for i, c in enumerate(calls):
        args, kwargs = [dec(a) for a in c["args"]], {k: dec(x) for k, x in c["kwargs"].items()}
        t = c["call"]
        if before is not None:
            before(i, t, args, kwargs)
        if isinstance(t, str):
            fn = substitute(t) if substitute is not None else None
            if fn is None:
                fn = _resolve(mod, t)
        else:
            fn = getattr(results[t["ref"]], t["method"])
        results.append(fn(*args, **kwargs))
    return results
