#!/usr/bin/env python3
"""Derive this project's parameter tables for the robots that share Walker3D's tree (Child3D, Mike) from the
reference MJCF files, and for the planar Walker2D / Crab2D (build container only).  Output: mocca_envs_amd/mjcf_tables.py -- numbers only, in the
Body / Hinge / Geom description format of mocca_envs_amd/model.py.  Gains come from robots.py, not from the XML."""
import os
import xml.etree.ElementTree as ET

REF = "/root/reference/mocca_envs/data/robots"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mocca_envs_amd", "mjcf_tables.py")


def f(s):
    return tuple(float(x) for x in s.split())


def body_src(b, gains, dgeom, ind):
    pad = "    " * ind
    joints = b.findall("joint")
    anchor = f(joints[0].get("pos", "0 0 0")) if joints else (0.0, 0.0, 0.0)
    for j in joints:
        assert f(j.get("pos", "0 0 0")) == anchor, "co-located hinges expected"
    out = [f"{pad}Body({b.get('name')!r}, {f(b.get('pos', '0 0 0'))!r}, anchor={anchor!r}, "
           f"quat_wxyz={f(b.get('quat', '1 0 0 0'))!r},"]
    out.append(f"{pad}     hinges=[")
    for j in joints:
        lo, hi = f(j.get("range"))
        out.append(f"{pad}         Hinge({j.get('name')!r}, {f(j.get('axis'))!r}, {lo!r}, {hi!r}, {gains[j.get('name')]!r}),")
    out.append(f"{pad}     ],")
    out.append(f"{pad}     geoms=[")
    for g in b.findall("geom"):
        r = f(g.get("size"))[0]
        grp = int(g.get("contype", dgeom["contype"]))
        msk = int(g.get("conaffinity", dgeom["conaffinity"]))
        fr = f(g.get("friction", dgeom["friction"]))[0]
        if g.get("type") == "capsule":
            ft = f(g.get("fromto"))
            out.append(f"{pad}         Geom({g.get('name')!r}, GEOM_CAPSULE, {r!r}, {ft[:3]!r}, {ft[3:]!r}, {grp}, {msk}, {fr!r}),")
        else:
            out.append(f"{pad}         Geom({g.get('name')!r}, GEOM_SPHERE, {r!r}, {f(g.get('pos', '0 0 0'))!r}, None, {grp}, {msk}, {fr!r}),")
    out.append(f"{pad}     ],")
    out.append(f"{pad}     children=[")
    for c in b.findall("body"):
        out.extend(body_src(c, gains, dgeom, ind + 2))
    out.append(f"{pad}     ]),")
    return out


def sub(a, b):
    return tuple(x - y for x, y in zip(a, b))


def body_src_global(b, origin_parent, gains, dgeom, ind, is_root=False):
    """<compiler coordinate="global">: every position in the file is a world position of the rest pose.  Body frames are
    ours to choose: the root's sits at the centre of its first geom (= the COM PyBullet reports for that link), a
    hinged body's at its joint.  The root's slide/slide/hinge "ignore*" joints (robots.py:163-165) are the planar
    freedom of the base and are not emitted: the kernel's floating base stays in the plane by symmetry."""
    pad = "    " * ind
    joints = [j for j in b.findall("joint") if not j.get("name").startswith("ignore")]
    if is_root:
        g0 = b.findall("geom")[0]
        if g0.get("type") == "capsule":
            ft = f(g0.get("fromto"))
            origin = tuple(0.5 * (ft[i] + ft[3 + i]) for i in range(3))
        else:
            origin = f(g0.get("pos"))
        pos = origin
    else:
        assert len(joints) == 1
        origin = f(joints[0].get("pos"))
        pos = sub(origin, origin_parent)
    out = [f"{pad}Body({b.get('name')!r}, {pos!r}, anchor=(0.0, 0.0, 0.0), quat_wxyz=(1.0, 0.0, 0.0, 0.0),"]
    out.append(f"{pad}     hinges=[")
    for j in joints:
        lo, hi = f(j.get("range"))
        out.append(f"{pad}         Hinge({j.get('name')!r}, {f(j.get('axis'))!r}, {lo!r}, {hi!r}, {gains[j.get('name')]!r}),")
    out.append(f"{pad}     ],")
    out.append(f"{pad}     geoms=[")
    for g in b.findall("geom"):
        r = f(g.get("size"))[0]
        grp = int(g.get("contype", dgeom["contype"]))
        msk = int(g.get("conaffinity", dgeom["conaffinity"]))
        fr = f(g.get("friction", dgeom["friction"]))[0]
        if g.get("type") == "capsule":
            ft = f(g.get("fromto"))
            out.append(f"{pad}         Geom({g.get('name')!r}, GEOM_CAPSULE, {r!r}, {sub(ft[:3], origin)!r}, {sub(ft[3:], origin)!r}, {grp}, {msk}, {fr!r}),")
        else:
            out.append(f"{pad}         Geom({g.get('name')!r}, GEOM_SPHERE, {r!r}, {sub(f(g.get('pos')), origin)!r}, None, {grp}, {msk}, {fr!r}),")
    out.append(f"{pad}     ],")
    out.append(f"{pad}     children=[")
    for c in b.findall("body"):
        out.extend(body_src_global(c, origin, gains, dgeom, ind + 2))
    out.append(f"{pad}     ]),")
    return out


WALKER2D_POWER = {"torso_joint": 100, "thigh_joint": 100, "leg_joint": 100, "foot_joint": 50, "thigh_left_joint": 100,
                  "leg_left_joint": 100, "foot_left_joint": 50}                                          # robots.py:342-350
CRAB2D_POWER = {"thigh_left_joint": 100, "leg_left_joint": 100, "foot_left_joint": 50, "thigh_joint": 100, "leg_joint": 100,
                "foot_joint": 50}                                                                         # robots.py:376-383
WALKER_POWER = {"abdomen_z": 60, "abdomen_y": 80, "abdomen_x": 60, "right_hip_x": 80, "right_hip_z": 60, "right_hip_y": 100,
                "right_knee": 90, "right_ankle": 60, "left_hip_x": 80, "left_hip_z": 60, "left_hip_y": 100, "left_knee": 90,
                "left_ankle": 60, "right_shoulder_x": 60, "right_shoulder_z": 60, "right_shoulder_y": 50, "right_elbow": 60,
                "left_shoulder_x": 60, "left_shoulder_z": 60, "left_shoulder_y": 50, "left_elbow": 60}   # robots.py:234-256
MIKE_POWER = dict(WALKER_POWER, abdomen_z=0, abdomen_y=0, abdomen_x=0, right_shoulder_x=30, right_shoulder_z=30,
                  right_shoulder_y=25, right_elbow=30, left_shoulder_x=30, left_shoulder_z=30, left_shoulder_y=25,
                  left_elbow=30)                                                                          # robots.py:477-499


def main():
    src = ['"""GENERATED by tools/gen_mjcf_tables.py from the reference\'s child3d.xml / mike.xml (numbers only, this',
           'project\'s description format).  Gains: robots.py:234-256 x base_power 0.4 (Child3D, :326-328), robots.py:477-499 (Mike)."""',
           "from .model import Body, Geom, Hinge, GEOM_CAPSULE, GEOM_SPHERE", "", ""]
    for name, xml, gains in (("child3d", "child3d.xml", {k: 0.4 * v for k, v in WALKER_POWER.items()}),
                             ("mike", "mike.xml", MIKE_POWER)):
        root = ET.parse(os.path.join(REF, xml)).getroot()
        d = root.find("default")
        dj, dg = d.find("joint").attrib, d.find("geom").attrib
        assert float(dj["armature"]) == 0.01 and float(dj["damping"]) == 0.1
        src.append(f"def {name}_description():")
        lines = body_src(root.find("worldbody").find("body"), gains, dg, 1)
        lines[0] = "    return " + lines[0].lstrip()
        lines[-1] = lines[-1].rstrip(",")
        src.extend(lines)
        src.extend(["", ""])
    for name, xml, gains in (("walker2d", "walker2d.xml", WALKER2D_POWER), ("crab2d", "crab2d.xml", CRAB2D_POWER)):
        root = ET.parse(os.path.join(REF, xml)).getroot()
        assert root.find("compiler").get("coordinate") == "global"
        d = root.find("default")
        dj, dg = d.find("joint").attrib, d.find("geom").attrib
        src.append(f"def {name}_description():")
        src.append(f'    """{xml}: joint armature {float(dj["armature"])!r}, damping {float(dj["damping"])!r} (defaults of the file)."""')
        lines = body_src_global(root.find("worldbody").find("body"), (0.0, 0.0, 0.0), gains, dg, 1, is_root=True)
        lines[0] = "    return " + lines[0].lstrip()
        lines[-1] = lines[-1].rstrip(",")
        src.extend(lines)
        src.extend(["", ""])
    open(OUT, "w").write("\n".join(src))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
