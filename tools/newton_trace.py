"""Developer tool: dump the first Newton Hessian / gradient / direction of env 0 and compare with the oracle."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gym-genesis_amd"), os.path.join(ROOT, "oracle")]
import orc
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
np.set_printoptions(linewidth=250, precision=4, suppress=False)
B = 4
spec = models.franka_cube_pick_scene().build()
sc = MirScene(spec, B); o = orc.Oracle(spec, B)
rng = np.random.RandomState(0)
pos = np.stack([rng.uniform(.45,.8,B), rng.uniform(-.25,.25,B), np.full(B,.02)],1).astype(np.float32)
quat = np.tile(np.array([0,0,0,1],np.float32),(B,1)); arm = np.tile(np.array(models.FRANKA_HOME,np.float32),(B,1))
sc.reset(pos,quat,arm); o.reset(pos,quat,arm)
acts = np.random.default_rng(1).uniform(-1,1,(40,B,9)).astype(np.float32)
sc.lib.mir_debug_profile_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
for t in range(9):
    sc.set_pd_targets(acts[t]); o.set_targets(acts[t])
    # teacher force so both start the step from the same state
    qo, vo = o.state(); ws = np.stack([o.read(orc.F_QACC_WS, e) for e in range(B)])
    sc.set_state(qpos=qo.astype(np.float32), qvel=vo.astype(np.float32), warmstart=ws.astype(np.float32))
    prof = torch.zeros(16 + 64 + 512, dtype=torch.int64, device=sc.device)
    sc._check(sc.lib.mir_debug_profile_step(sc.h, C.c_void_p(prof.data_ptr()), sc._stream()))
    o.step_batch(None)
    dbg = prof[16 + 64:].cpu().numpy().view(np.float32)
    H = dbg[:256].reshape(16, 16)[:15, :15].astype(np.float64); g = dbg[256:271].astype(np.float64); s = dbg[272:287].astype(np.float64)
    nc, ne, ni = (x.cpu().numpy() for x in sc.get_diag())
    if t < 7: continue
    Mt = o.read(orc.F_MT, 0).reshape(15, 15); J = o.read(orc.F_J, 0).reshape(-1, 15); D = o.read(orc.F_EFCD, 0); aref = o.read(orc.F_AREF, 0)
    qas = o.read(orc.F_QACC_SMOOTH, 0)
    for name, a0 in (("ws", ws[0]), ("smooth", qas)):
        jar = J @ a0 - aref; act = jar < 0
        Ho = Mt + J.T @ np.diag(D * act) @ J
        print(f"step {t} start={name}: niter hip {ni[0]} orc {o.counts(0)}; |H-Ho|max {np.abs(H-Ho).max():.3e} (|Ho|max {np.abs(Ho).max():.3e})")
        go = Mt @ a0 - o.read(orc.F_QFRC_SMOOTH, 0) - J.T @ (-(D * act) * jar)
        print("   |g-go|", np.abs(g - go).max(), " |s - solve(H,-g)|", np.abs(s - np.linalg.solve(H, -g)).max(), "|s|", np.abs(s).max())
    jar_h = dbg[288:288+64].reshape(16,4); ljar_h = dbg[352:368]; lact_h = dbg[368:384]; lsg_h = dbg[384:400]; bits_h = dbg[400:416]
    jo = J @ qas - aref
    nlim = J.shape[0] - 4 * int(nc[0])
    print("   hip jar (contact x row):\n", jar_h[:int(nc[0])], "\n   bits", bits_h[:int(nc[0])])
    print("   orc jar contacts:\n", jo[nlim:].reshape(-1, 4))
    print("   hip lsg", lsg_h[:9], "\n   hip ljar", ljar_h[:9], "\n   hip lact", lact_h[:9])
    print("   orc limit rows: dof", [int(np.argmax(np.abs(J[r]))) for r in range(nlim)], "jar", jo[:nlim], "D", D[:nlim])
    H0 = dbg[416:416+256].reshape(16,16)[:15,:15].astype(np.float64)
    print("   pre-contact |H0 - (Mt + limits)|:", np.abs(H0 - (Mt + J[:nlim].T @ np.diag(D[:nlim] * act[:nlim]) @ J[:nlim])).max(), " H0 diag", np.diag(H0)[6:9], "Mt diag", np.diag(Mt)[6:9])
    print("   |H0 - H0^T|", np.abs(H0 - H0.T).max(), "|H - H^T|", np.abs(H - H.T).max())
    d = H - Ho
    print("   diff rows 9..14 (cube block):\n", d[9:, 9:])
    print("   Ho cube block:\n", Ho[9:, 9:])
    print("   diff arm diag", np.diag(d)[:9])
    print("   H cube block:\n", H[9:, 9:])
    Jc = J[nlim:]; print("   orc J contact rows (cube cols):\n", Jc[:, 9:])
