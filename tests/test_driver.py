"""The yaml-driven dycore-only driver: config parsing on CPU, a short run on the GPU."""
import json
import os

import pytest

from pace_amd import driver

YAML = """
dycore_only: true
disable_step_physics: true
stencil_config:
  compilation_config:
    backend: gt:gpu
    device_sync: false
initialization:
  type: analytic
  config:
    case: baroclinic
performance_config:
  collect_performance: true
  experiment_name: c12_test
nx_tile: 12
nz: 79
dt_atmos: 225
minutes: 15
layout: [1, 1]
dycore_config:
  a_imp: 1.0
  beta: 0.
  d4_bg: 0.15
  hord_dp: 6
  hord_tr: 8
  k_split: 1
  n_split: 2
  nord: 3
  kord_tm: -9
  n_sponge: 48
"""


def test_yaml_is_mapped_like_the_reference_driver(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text(YAML)
    run, dy, ignored = driver.load_config(str(p))
    assert run["nx_tile"] == 12 and run["nz"] == 79 and run["layout"] == (1, 1)
    assert run["n_steps"] == 4  # 15 minutes / 225 s  [REF driver.py: total_time / dt_atmos]
    assert run["dycore_only"] and run["disable_step_physics"] and run["backend"] == "gt:gpu"
    assert dy["n_split"] == 2 and dy["nord"] == 3 and dy["d4_bg"] == 0.15
    assert "kord_tm" in ignored and "hord_dp" not in ignored  # remap options: outside the acoustic path


@pytest.mark.gpu
def test_driver_runs_the_reference_c12_config_shape(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text(YAML)
    out = tmp_path / "perf.json"
    assert driver.main([str(p), "--steps", "3", "--out", str(out)]) == 0
    d = json.load(open(out))
    assert "mainloop" not in d["times"] and d["setup"]["acoustic_only"]  # not the reference's full step_dynamics timer
    times = d["times"]["acoustic_mainloop"]["times"]
    assert len(times) == 6 and all(len(t) == 3 for t in times)  # one entry per rank and step, as the reference collector
    assert d["setup"]["finite"] and d["acoustic_simulated_days_per_day"] > 0
    dc = d["times"]["DynCore"]  # the dycore's own clock around the acoustic dynamics [REF tests/main/driver/test_driver.py:81-85]
    assert dc["hits"] == 3 and all(0 < a <= b for ta, tb in zip(dc["times"], times) for a, b in zip(ta, tb))


@pytest.mark.gpu
def test_driver_runs_the_step_dynamics_body(tmp_path):
    """--tracers N --remap: k_split x [acoustic call, tracer advection, vertical remap]; the json says so."""
    p = tmp_path / "c.yaml"
    p.write_text(YAML)
    out = tmp_path / "perf.json"
    assert driver.main([str(p), "--steps", "3", "--tracers", "2", "--remap", "--out", str(out)]) == 0
    d = json.load(open(out))
    assert not d["setup"]["acoustic_only"] and d["setup"]["tracers"] == 2 and d["setup"]["remap"] and d["setup"]["finite"]
    # the reference's timer names, and its own performance script's arithmetic on the file [REF .jenkins/print_performance_number.py:11-15]
    import numpy as np

    assert set(d["times"]) == {"mainloop", "DynCore", "TracerAdvection", "Remapping"}
    for rank in range(6):
        m = np.mean(d["times"]["mainloop"]["times"][rank][1:])
        parts = sum(np.mean(d["times"][n]["times"][rank][1:]) for n in ("DynCore", "TracerAdvection", "Remapping"))
        assert 0 < parts <= m
    assert d["times"]["mainloop"]["hits"] == 3



def test_timer_has_the_reference_interface():
    """ndsl.performance.timer.Timer as the reference driver uses it [REF driver/pace/driver/driver.py:630-643]."""
    from pace_amd.timer import NullTimer, Timer

    n = []
    t = Timer(sync=lambda: n.append(1))
    with t.clock("mainloop"):
        with t.clock("DynCore"):
            pass
        with t.clock("DynCore"):
            pass
    assert t.hits == {"DynCore": 2, "mainloop": 1} and t.times["mainloop"] >= t.times["DynCore"] >= 0 and len(n) == 6
    t.start("x")
    with pytest.raises(RuntimeError):
        t.times
    with pytest.raises(ValueError):
        t.start("x")
    t.stop("x")
    t.reset()
    assert t.times == {} and t.hits == {}
    z = NullTimer()
    with z.clock("a"):
        pass
    assert z.times == {} and not z.enabled
