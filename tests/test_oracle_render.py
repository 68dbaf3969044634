"""Known-answer tests that pin the float64 ray-cast oracle of the `pixels` path (oracle/orc_render.c).

The reference's images come from Genesis's rasteriser over mesh assets that are not in /root/reference, so there
are no golden images to compare with (parity unpinned); what CAN be pinned is the camera contract the
reference states (res=(W,H), vertical fov, pos/lookat, uint8 HxWx3, row 0 = top: cube_pick.py:41,56-63) and the image
definition of include/mirigid.h, against closed-form projections of a hand-built scene.
"""
import math

import numpy as np

import orc
from gym_genesis.backend.spec import GEOM_BOX, GEOM_PLANE, JNT_FREE, SceneBuilder, box_inertia, make_camera


def _scene(box_pos=(0.0, 0.0, 0.5), half=(0.1, 0.2, 0.5), rgb=(1.0, 0.5, 0.25)):
    sb = SceneBuilder()
    sb.add_geom(0, GEOM_PLANE)
    sb.add_body("box", 0, pos=box_pos, jtype=JNT_FREE, mass=1.0, inertia=box_inertia(1.0, half))
    sb.add_geom("box", GEOM_BOX, size=half, rgb=rgb)
    xpos = np.array([[[0, 0, 0], box_pos]], dtype=np.float64)
    xquat = np.array([[[1, 0, 0, 0], [1, 0, 0, 0]]], dtype=np.float64)
    return sb, xpos, xquat


def _u8(c):
    return int(math.floor(min(max(c, 0.0), 1.0) * 255.0 + 0.5))


def test_top_down_box_silhouette_and_shading():
    sb, xpos, xquat = _scene()
    W, H, fov, cz = 200, 100, 60.0, 3.0
    cam = make_camera(W, H, (0, 0, cz), (0, 0, 0), fov, up=(0, 1, 0))  # looking straight down, +y up in the image
    vis = sb.visual(light_dir=(0.0, 0.0, 1.0), ambient=0.25, diffuse=0.5)
    img, depth = orc.render_image(sb.build(), cam, vis, xpos, xquat, want_depth=True)
    assert img.shape == (H, W, 3) and img.dtype == np.uint8
    # box top at z=1 is lit head-on: albedo * (0.25 + 0.5)
    box_rgb = tuple(_u8(c * 0.75) for c in (1.0, 0.5, 0.25))
    assert tuple(img[H // 2, W // 2]) == box_rgb
    # closed-form silhouette of the top face (the nearest part of the box): |x| <= 0.1, |y| <= 0.2 at depth cz - 1
    ty = math.tan(math.radians(fov) / 2)
    tx = ty * W / H
    is_box = np.all(img == np.array(box_rgb, np.uint8), axis=-1)
    cols = np.where(is_box[H // 2])[0]
    rows = np.where(is_box[:, W // 2])[0]
    px_per_m_x = (W / 2) / (tx * (cz - 1.0))
    px_per_m_y = (H / 2) / (ty * (cz - 1.0))
    assert abs(len(cols) - 2 * 0.1 * px_per_m_x) <= 1.0
    assert abs(len(rows) - 2 * 0.2 * px_per_m_y) <= 1.0
    assert abs(cols.mean() - (W - 1) / 2) <= 0.5 and abs(rows.mean() - (H - 1) / 2) <= 0.5
    # ray parameter of the centre pixel: forward is unit length, so t = distance along the axis
    assert abs(depth[H // 2, W // 2] - (cz - 1.0)) < 1e-3 * cz and abs(depth[0, 0] - cz) / cz < 0.5


def test_row_zero_is_the_top_and_sky_above_the_horizon():
    sb, xpos, xquat = _scene(box_pos=(50.0, 0.0, 0.5))  # box out of view
    cam = make_camera(64, 48, (0, 0, 1.0), (1.0, 0, 1.0), 60.0)  # horizontal view: horizon through the middle
    vis = sb.visual()
    img = orc.render_image(sb.build(), cam, vis, xpos, xquat)
    sky = tuple(_u8(c) for c in (0.55, 0.7, 0.9))
    assert all(tuple(img[0, i]) == sky for i in range(64))       # top rows: sky
    assert tuple(img[47, 32]) != sky                               # bottom rows: ground
    assert np.all(np.all(img[:24] == np.array(sky, np.uint8), axis=-1)) and not np.any(np.all(img[24:] == np.array(sky, np.uint8), axis=-1))


def test_checker_cells_and_plane_shading():
    sb, xpos, xquat = _scene(box_pos=(50.0, 0.0, 0.5))
    W = H = 101
    cam = make_camera(W, H, (0.25, 0.25, 2.0), (0.25, 0.25, 0.0), 40.0, up=(0, 1, 0))  # above the centre of cell (0, 0)
    vis = sb.visual(light_dir=(0, 0, 1), ambient=0.2, diffuse=0.6, checker_rgb=((1.0, 1.0, 1.0), (0.5, 0.0, 0.0)), checker_size=0.5)
    img = orc.render_image(sb.build(), cam, vis, xpos, xquat)
    even, odd = (_u8(0.8),) * 3, (_u8(0.4), 0, 0)
    assert tuple(img[H // 2, W // 2]) == even      # cell (0,0): parity 0
    # one cell to the +x side of the image centre (image right = world +x for this camera): parity 1
    px_per_m = (W / 2) / (math.tan(math.radians(20.0)) * 2.0)
    dx = int(round(0.5 * px_per_m))
    assert tuple(img[H // 2, W // 2 + dx]) == odd and tuple(img[H // 2 - dx, W // 2]) == odd
    assert tuple(img[H // 2 - dx, W // 2 + dx]) == even
    assert set(map(tuple, img.reshape(-1, 3))) == {even, odd}


def test_nearest_surface_wins_and_side_faces_use_their_own_normals():
    sb, xpos, xquat = _scene(box_pos=(0.0, 0.0, 0.5), half=(0.5, 0.5, 0.5), rgb=(1.0, 1.0, 1.0))
    cam = make_camera(120, 90, (4.0, 0.0, 0.5), (0.0, 0.0, 0.5), 30.0)  # looking along -x at the +x face
    lx = (0.6, 0.0, 0.8)
    vis = sb.visual(light_dir=lx, ambient=0.1, diffuse=0.9)
    img = orc.render_image(sb.build(), cam, vis, xpos, xquat)
    assert tuple(img[45, 60]) == (_u8(0.1 + 0.9 * 0.6),) * 3  # +x face: n.l = 0.6
    # with the light behind the face only the ambient term remains
    vis2 = sb.visual(light_dir=(-1.0, 0.0, 0.0), ambient=0.1, diffuse=0.9)
    img2 = orc.render_image(sb.build(), cam, vis2, xpos, xquat)
    assert tuple(img2[45, 60]) == (_u8(0.1),) * 3


def test_global_image_offsets_and_single_plane():
    """Two copies of the env displaced by env offsets appear mirrored about the image centre; the plane is drawn once."""
    sb, xpos, xquat = _scene(box_pos=(0.0, 0.0, 0.2), half=(0.2, 0.2, 0.2), rgb=(0.0, 1.0, 0.0))
    cam = make_camera(160, 80, (0, 0, 6.0), (0, 0, 0), 50.0, up=(0, 1, 0))
    vis = sb.visual(light_dir=(0, 0, 1), ambient=0.5, diffuse=0.5, checker_rgb=((0.3, 0.3, 0.3), (0.3, 0.3, 0.3)))
    xp2, xq2 = np.repeat(xpos, 2, 0), np.repeat(xquat, 2, 0)
    off = np.array([[-1.0, 0, 0], [1.0, 0, 0]])
    img = orc.render_image(sb.build(), cam, vis, xp2, xq2, offsets=off)
    green = np.all(img == np.array([0, 255, 0], np.uint8), axis=-1)
    cols = np.where(green[40])[0]
    assert len(cols) > 0 and abs(cols.mean() - 79.5) < 0.51          # symmetric pair
    gaps = np.where(np.diff(cols) > 1)[0]
    assert len(gaps) == 1                                              # exactly two separate boxes
    assert np.array_equal(green, green[:, ::-1])
