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
                 speed=10.0, dt=0.25, ccw=False, noise_scale=7.0, rows=ROWS, cols=COLS,
                 bump_sigma=1.5, skirt=0.0, skirt_decay=12.0, occlusion=0.6, circle_frames=0, near_clutter=True):
        self.seed = seed
        self.range_res = np.float32(range_res)
        self.rows, self.cols = rows, cols
        self.speed, self.dt, self.ccw = speed, dt, ccw
        self.noise_scale = noise_scale
        # dense variant (scene_dense): wider returns with range side-lobe skirts and weaker occlusion
        self.bump_sigma, self.skirt, self.skirt_decay, self.occlusion = bump_sigma, skirt, skirt_decay, occlusion
        # circle_frames = F > 0: the sensor drives a closed circle of F frames (constant yaw rate 2 pi / (F dt)),
        # so frame F continues frame F - 1 into frame 0: a ring of F frames is an endless sequence
        self.circle_frames = circle_frames
        self.near_clutter = near_clutter
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
        if self.circle_frames:
            w = np.full(nsteps, 2 * np.pi / (self.circle_frames * self.dt))
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
        amp = self.wall_amp[wi] * (self.occlusion ** rank[ai, wi])
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
        sig = self.bump_sigma
        reach = 5 if (sig == 1.5 and self.skirt == 0.0) else int(max(np.ceil(3.5 * sig), 3 * self.skirt_decay * (self.skirt > 0)))
        for off in range(-reach, reach + 1):
            bi = np.round(b).astype(int) + off
            m = (bi >= 0) & (bi < cols)
            d = bi[m] - b[m]
            v = np.exp(-(d ** 2) / (2 * sig ** 2))
            if self.skirt > 0:
                v = v + self.skirt * np.exp(-np.abs(d) / self.skirt_decay)
            np.add.at(img, (ha[m], bi[m]), hamp[m] * v)
        # near-field clutter: the first bins are strong (consume k-strongest slots, then dropped
        # by min_distance -- radar_filters.cpp:315,327)
        near = rng.uniform(100.0, 255.0, size=(rows, 10))      # drawn in every variant: the stream of later draws stays put
        if self.near_clutter:
            img[:, :10] = near
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


DENSE_KW = dict(n_walls=90, noise_scale=12.5, bump_sigma=2.5, skirt=0.35, skirt_decay=10.0, occlusion=0.8, near_clutter=False)


def scene_dense(seed, n_frames, **kw):
    """Dense-row variant of scene_v1: every azimuth holds at least k = 40 bins >= z_min = 60 (wide returns with
    range side lobes, a higher noise floor), so the k-strongest filter cuts every row (N_f ~ rows * k, the
    reference's upper bound, radar_filters.cpp:214-229) and no row takes the "few candidates" shortcut."""
    a = dict(DENSE_KW)
    a.update(kw)
    return scene_v1(seed, n_frames, **a)


def uniform_v1(seed, rows=ROWS, cols=COLS, batch=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(batch, rows, cols), dtype=np.uint8)


def _hits_all_frames(sc, frames, device):
    """Ray / scene intersections of `frames` sweeps at once, in torch on `device` -> (flat row index f * rows + a,
    range [m], amplitude): the same geometry as Scene.render (walls with occlusion ranks, point scatterers inside
    the half beam)."""
    import torch
    rows, cols = sc.rows, sc.cols
    F = len(frames)
    n_frames = int(max(frames)) + 1
    X, Y, TH = sc._integrate(n_frames)
    a = np.arange(rows)
    theta = (a + 1) / rows * 2 * np.pi
    dfrac = theta / (2 * np.pi) - 0.5
    if sc.ccw:
        dfrac = -dfrac
    fr = np.asarray(frames, np.float64)[:, None]
    ti = np.clip(np.round((fr + 0.5 + dfrac[None, :]) * rows).astype(np.int64), 0, X.size - 1).reshape(-1)   # [F * rows]
    T = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(device)
    ox, oy, oth = T(X[ti]), T(Y[ti]), T(TH[ti])
    th = oth + T(np.tile(theta, F))
    ux, uy = torch.cos(th), torch.sin(th)
    res = float(sc.range_res)
    ex, ey = T((sc.Q - sc.P)[:, 0]), T((sc.Q - sc.P)[:, 1])
    px, py = T(sc.P[:, 0])[None, :] - ox[:, None], T(sc.P[:, 1])[None, :] - oy[:, None]
    den = ux[:, None] * ey[None, :] - uy[:, None] * ex[None, :]
    rho = (px * ey[None, :] - py * ex[None, :]) / den
    s = (px * uy[:, None] - py * ux[:, None]) / den
    ok = (den.abs() > 1e-9) & (rho > 0.5) & (s >= 0) & (s <= 1) & (rho < res * cols)
    rho_m = torch.where(ok, rho, torch.full_like(rho, float("inf")))
    rank = torch.argsort(torch.argsort(rho_m, dim=1), dim=1)
    ai, wi = torch.nonzero(ok, as_tuple=True)
    amp = T(sc.wall_amp)[wi] * (sc.occlusion ** rank[ai, wi].to(torch.float64))
    sx, sy = T(sc.S[:, 0])[None, :] - ox[:, None], T(sc.S[:, 1])[None, :] - oy[:, None]
    along = sx * ux[:, None] + sy * uy[:, None]
    perp = (sx * uy[:, None] - sy * ux[:, None]).abs()
    oks = (along > 0.5) & (along < res * cols) & (perp < along * float(np.tan(np.pi / rows)))
    aj, si = torch.nonzero(oks, as_tuple=True)
    ha = torch.cat([ai, aj])
    hr = torch.cat([rho[ai, wi], along[aj, si]])
    hamp = torch.cat([amp, T(sc.scat_amp)[si]])
    return ha, hr, hamp


def render_frames_torch(sc, frames, device, generator=None):
    """Renders the sweeps `frames` of Scene `sc` on the GPU -> uint8 torch tensor [len(frames), rows, cols].  Geometry
    as Scene.render (NumPy, all frames at once); noise, speckle, return shapes, clutter and streaks are drawn with
    torch's generator, so the images are a different random draw than Scene.render's (bench input, not a fixture)."""
    import torch
    rows, cols = sc.rows, sc.cols
    F = len(frames)
    g = generator
    if g is None:
        g = torch.Generator(device=device)
        g.manual_seed(1_000_003 * int(sc.seed) + 17)
    ha_t, hr, hamp = _hits_all_frames(sc, frames, device)
    res = float(sc.range_res)
    img = torch.empty((F * rows, cols), dtype=torch.float32, device=device).exponential_(1.0 / sc.noise_scale, generator=g) + 10.0
    hr_t = hr.to(torch.float32)
    amp_t = hamp.to(torch.float32)
    amp_t = amp_t * torch.clamp(60.0 / hr_t, max=1.0) * (0.7 + 0.4 * torch.rand(amp_t.shape, device=device, generator=g))
    b = (hr_t - res / 2) / res
    bi0 = torch.round(b).to(torch.int64)
    sig = sc.bump_sigma
    reach = 5 if (sig == 1.5 and sc.skirt == 0.0) else int(max(np.ceil(3.5 * sig), 3 * sc.skirt_decay * (sc.skirt > 0)))
    flat = img.view(-1)
    for off in range(-reach, reach + 1):
        bi = bi0 + off
        m = (bi >= 0) & (bi < cols)
        d = bi.to(torch.float32) - b
        v = torch.exp(-(d * d) / (2 * sig * sig))
        if sc.skirt > 0:
            v = v + sc.skirt * torch.exp(-torch.abs(d) / sc.skirt_decay)
        flat.index_add_(0, (ha_t * cols + bi)[m], (amp_t * v)[m])
    if sc.near_clutter:
        img[:, :10] = 100.0 + 155.0 * torch.rand((F * rows, 10), device=device, generator=g)
    streak = torch.nonzero(torch.rand(F * rows, device=device, generator=g) < 0.05).view(-1).cpu().numpy()
    rs = np.random.Generator(np.random.PCG64([int(sc.seed), 104729, int(frames[0])]))
    for r in streak:
        start = int(rs.integers(200, cols - 100))
        img[int(r), start:start + int(rs.integers(20, 60))] = 255.0
    return torch.clamp(torch.round(img), 0, 255).to(torch.uint8).view(F, rows, cols)
