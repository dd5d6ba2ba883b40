"""samples the GPU's shader clock / power / temperature from sysfs (hwmon) in a background thread while a workload runs;
used to tell a throttled box from a scheduling problem when the same job times differently on two boxes.
usage (module): with ClockWatch() as cw: run(); print(cw.summary())
usage (script): python tools/clock_watch.py [X,Y,Z] [batch]   -> whole-volume inference three times, with the samples"""
import glob
import json
import os
import sys
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return None


class ClockWatch(object):
    def __init__(self, period=0.02):
        self.period = period
        self.samples = []
        self._stop = threading.Event()
        self.files = {}
        for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
            for key, name in (('sclk_hz', 'freq1_input'), ('power_uw', 'power1_average'), ('power_in_uw', 'power1_input'),
                              ('temp_mC', 'temp1_input'), ('temp_hot_mC', 'temp2_input')):
                p = os.path.join(hw, name)
                if os.path.exists(p) and key not in self.files:
                    self.files[key] = p
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            row = {'t': time.time()}
            for k, p in self.files.items():
                v = _read(p)
                try:
                    row[k] = float(v)
                except Exception:
                    pass
            self.samples.append(row)
            time.sleep(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join()

    def summary(self):
        out = {'n': len(self.samples), 'files': self.files}
        for k in self.files:
            vals = [s[k] for s in self.samples if k in s]
            if vals:
                out[k] = {'min': min(vals), 'max': max(vals), 'mean': sum(vals) / len(vals)}
        return out


if __name__ == '__main__':
    import torch
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd'))
    sys.path.insert(0, REPO)
    import bench
    from segmentation3d.network import vnet
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = vnet.SegmentationNet(1, 2)
    vnet.parameters_kaiming_init(net)
    net = net.to(dev).eval()
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    vol = tuple(int(v) for v in (argv[0] if len(argv) > 0 else '512,512,400').split(','))
    batch = int(argv[1]) if len(argv) > 1 else 16
    with ClockWatch() as cw:
        r = bench.time_inference(net, vol, 96, 48, 2, batch, dev, two_streams='--single-stream' not in sys.argv)
    print(json.dumps({'seconds_all_runs': r['seconds_all_runs'], 'clock_watch': cw.summary()}))
