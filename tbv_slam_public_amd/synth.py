"""Synthetic polar radar sweeps (there is no radar data in the reference repository; the bags live
on Google Drive, /root/reference/README.md:34-36).  Generators follow SURVEY.md section 8(d):

  scene_v1(seed)   -- piecewise-straight walls + point scatterers seen from a sensor driving along
                      arcs at 10 m/s, 4 Hz, 400 azimuths x 3360 range bins, uint8; the sweep is
                      rendered azimuth by azimuth from the interpolated sensor pose, so the motion
                      distortion that `Compensate` (utils.cpp:96-107) removes is really there.
  uniform_v1(seed) -- i.i.d. uniform uint8 image: worst-case ties for the bit-exact filter tests.

NumPy only (PCG64), deterministic for a given seed on every machine.
"""
import numpy as np

ROWS, COLS = 400, 3360


class Scene:
    def __init__(self, seed, n_walls=60, n_scatter=200, extent=300.0, range_res=0.0438,
                 speed=10.0, dt=0.25, ccw=False, noise_scale=7.0, rows=ROWS, cols=COLS):
        self.seed = seed
        self.range_res = np.float32(range_res)
        self.rows, self.cols = rows, cols
        self.speed, self.dt, self.ccw = speed, dt, ccw
        self.noise_scale = noise_scale
        rng = np.random.Generator(np.random.PCG64(seed))
        # scale the world with the sensor range so every preset sees a comparable number of walls
        max_range = float(range_res) * cols
        extent = extent * max(1.0, max_range / 150.0)
        c = rng.uniform(-extent / 2, extent / 2, size=(n_walls, 2))
        length = rng.uniform(10.0, 80.0, size=n_walls) * max(1.0, max_range / 150.0)
        ang = rng.uniform(0, np.pi, size=n_walls)
        d = np.stack([np.cos(ang), np.sin(ang)], 1) * length[:, None] / 2
        self.P, self.Q = c - d, c + d
        self.S = rng.uniform(-extent / 2, extent / 2, size=(n_scatter, 2))
        self.wall_amp = rng.uniform(120.0, 240.0, size=n_walls)
        self.scat_amp = rng.uniform(120.0, 240.0, size=n_scatter)
        # yaw-rate profile: piecewise constant, U[-5,5] deg/s, re-drawn every 5 s
        self._yaw_rates = np.deg2rad(rng.uniform(-5.0, 5.0, size=4096))
        self._traj_cache = {}

    # ---- trajectory: unicycle integrated at azimuth resolution -------------------------------
    def _integrate(self, n_frames):
        key = n_frames
        if key in self._traj_cache:
            return self._traj_cache[key]
        sub = self.rows
        h = self.dt / sub
        nsteps = (n_frames + 1) * sub
        seg = (np.arange(nsteps) * h / 5.0).astype(int)
        w = self._yaw_rates[seg % self._yaw_rates.size]
        th = np.concatenate([[0.0], np.cumsum(w * h)])[:-1]
        x = np.concatenate([[0.0], np.cumsum(self.speed * np.cos(th) * h)])[:-1]
        y = np.concatenate([[0.0], np.cumsum(self.speed * np.sin(th) * h)])[:-1]
        self._traj_cache[key] = (x, y, th)
        return x, y, th

    def pose_at(self, frame, n_frames=None):
        """Ground-truth sensor pose (x, y, theta) at the mid-sweep time of `frame`."""
        n_frames = max(frame + 1, n_frames or 0)
        x, y, th = self._integrate(n_frames)
        i = frame * self.rows + self.rows // 2
        return np.array([x[i], y[i], th[i]])

    # ---- rendering --------------------------------------------------------------------------
    def render(self, frame, n_frames=None):
        n_frames = max(frame + 1, n_frames or 0)
        rows, cols = self.rows, self.cols
        rng = np.random.Generator(np.random.PCG64([self.seed, 7919, frame]))
        X, Y, TH = self._integrate(n_frames)
        a = np.arange(rows)
        theta = (a + 1) / rows * 2 * np.pi                   # radar_filters.cpp:317
        # time fraction of azimuth a inside the sweep (utils.h:28-32): d = theta/2pi - 0.5 (cw)
        dfrac = theta / (2 * np.pi) - 0.5
        if self.ccw:
            dfrac = -dfrac
        ti = np.clip(np.round((frame + 0.5 + dfrac) * rows).astype(int), 0, X.size - 1)
        ox, oy, oth = X[ti], Y[ti], TH[ti]
        ux, uy = np.cos(oth + theta), np.sin(oth + theta)
        res = float(self.range_res)

        img = 10.0 + rng.exponential(self.noise_scale, size=(rows, cols))
        hits_a, hits_r, hits_amp = [], [], []
        # walls: ray/segment intersection, [rows, n_walls]
        ex, ey = (self.Q - self.P).T
        px, py = self.P[:, 0][None, :] - ox[:, None], self.P[:, 1][None, :] - oy[:, None]
        den = ux[:, None] * ey[None, :] - uy[:, None] * ex[None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            rho = (px * ey[None, :] - py * ex[None, :]) / den
            s = (px * uy[:, None] - py * ux[:, None]) / den
        ok = (np.abs(den) > 1e-9) & (rho > 0.5) & (s >= 0) & (s <= 1) & (rho < res * cols)
        # attenuation by occlusion: each nearer hit on the same ray halves the return
        rho_m = np.where(ok, rho, np.inf)
        order = np.argsort(rho_m, axis=1)
        rank = np.empty_like(order)
        np.put_along_axis(rank, order, np.arange(rho.shape[1])[None, :], axis=1)
        ai, wi = np.nonzero(ok)
        amp = self.wall_amp[wi] * (0.6 ** rank[ai, wi])
        hits_a.append(ai); hits_r.append(rho[ai, wi]); hits_amp.append(amp)
        # point scatterers: inside the half-beam
        sx, sy = self.S[:, 0][None, :] - ox[:, None], self.S[:, 1][None, :] - oy[:, None]
        along = sx * ux[:, None] + sy * uy[:, None]
        perp = np.abs(sx * uy[:, None] - sy * ux[:, None])
        oks = (along > 0.5) & (along < res * cols) & (perp < along * np.tan(np.pi / rows))
        ai, si = np.nonzero(oks)
        hits_a.append(ai); hits_r.append(along[ai, si]); hits_amp.append(self.scat_amp[si])
        ha = np.concatenate(hits_a); hr = np.concatenate(hits_r); hamp = np.concatenate(hits_amp)
        # 1/rho falloff beyond 60 m + multiplicative speckle
        hamp = hamp * np.minimum(1.0, 60.0 / hr) * rng.uniform(0.7, 1.1, size=hamp.shape)
        b = (hr - res / 2) / res
        for off in range(-5, 6):
            bi = np.round(b).astype(int) + off
            m = (bi >= 0) & (bi < cols)
            np.add.at(img, (ha[m], bi[m]), hamp[m] * np.exp(-((bi[m] - b[m]) ** 2) / (2 * 1.5 ** 2)))
        # near-field clutter: the first bins are strong (consume k-strongest slots, then dropped
        # by min_distance -- radar_filters.cpp:315,327)
        img[:, :10] = rng.uniform(100.0, 255.0, size=(rows, 10))
        # saturated streaks on 5 % of the azimuths: 255 plateaus (tie stress)
        streak = rng.random(rows) < 0.05
        for r in np.nonzero(streak)[0]:
            start = int(rng.integers(200, cols - 100))
            img[r, start:start + int(rng.integers(20, 60))] = 255.0
        return np.clip(np.round(img), 0, 255).astype(np.uint8)


def scene_v1(seed, n_frames, **kw):
    """Returns (images uint8 [n_frames, 400, 3360], gt poses [n_frames, 3], Scene)."""
    sc = Scene(seed, **kw)
    imgs = np.stack([sc.render(f, n_frames) for f in range(n_frames)])
    gt = np.stack([sc.pose_at(f, n_frames) for f in range(n_frames)])
    return imgs, gt, sc


def uniform_v1(seed, rows=ROWS, cols=COLS, batch=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(batch, rows, cols), dtype=np.uint8)
