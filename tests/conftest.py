import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


CONFIGS = {
    # name: (yaml relative to eagle-mpc_amd/data/yaml, dt in ms)  -- BASELINE.json configs 1-4
    "hover": ("hexacopter370/trajectories/hover.yaml", 40),
    "displacement": ("hexacopter370_flying_arm_3/trajectories/displacement.yaml", 80),
    "eagle_catch": ("hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", 32),
    "push_slide": ("hextilt_flying_arm_5/trajectories/push_slide.yaml", 13),
}


@pytest.fixture(scope="session")
def empc():
    import empc_loader
    mod = empc_loader.load()
    if not os.path.exists(mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return mod


@pytest.fixture(scope="session")
def problems(empc):
    """name -> (Trajectory, Problem)"""
    out = {}
    for name, (rel, dt) in CONFIGS.items():
        t = empc.Trajectory()
        t.autoSetup(empc.yaml_path(rel))
        out[name] = (t, t.createProblem(dt, True, "IntegratedActionModelEuler"))
    return out


def contact_variant(empc, tmp_path, contact="ContactModel3D", gains=(0.0, 0.0), dt=32, integrator="IntegratedActionModelEuler",
                    squash=True):
    """eagle_catch with its grasp-stage contact swapped for `contact` (ContactModel3D | ContactModel6D) and Baumgarte
    `gains` -- the two factory options of src/factory/contacts.cpp:26-79 that no shipped YAML exercises.  Returns
    (trajectory, problem)."""
    src = open(empc.yaml_path(CONFIGS["eagle_catch"][0])).read()
    old = '          type: "ContactModel3D"\n          link_name: "flying_arm_3__gripper"\n          position: [0, 0, 0]\n          gains: [0, 0]\n'
    assert src.count(old) == 1
    new = '          type: "%s"\n          link_name: "flying_arm_3__gripper"\n          position: [0, 0, 0]\n' % contact
    if contact == "ContactModel6D":
        new += '          orientation: [0, 0, 0, 1]\n'
    new += '          gains: [%r, %r]\n' % (float(gains[0]), float(gains[1]))
    f = tmp_path / ("eagle_catch_%s_%g_%g.yaml" % (contact, gains[0], gains[1]))
    f.write_text(src.replace(old, new))
    tr = empc.Trajectory()
    tr.autoSetup(str(f))
    return tr, tr.createProblem(dt, squash, integrator)


def two_contact_variant(empc, tmp_path, second="ContactModel3D", gains=(0.0, 0.0), gains2=(0.0, 0.0), dt=32,
                        integrator="IntegratedActionModelEuler", link2="flying_arm_3__link_2", squash=True, cone_on_second=False,
                        bent=None, name2="elbow", extra_frame_cost=None):
    """eagle_catch with a SECOND contact in its grasp stage (src/stage.cpp:38-48 adds every name of the stage's `contacts` list
    to one ContactModelMultiple; no shipped YAML lists more than one).  The new contact is called "elbow": crocoddyl's
    name-sorted map puts it BEFORE "end_effector", so its rows come first in the stacked Jacobian.  Returns (trajectory, problem)."""
    src = open(empc.yaml_path(CONFIGS["eagle_catch"][0])).read()
    old = '          type: "ContactModel3D"\n          link_name: "flying_arm_3__gripper"\n          position: [0, 0, 0]\n          gains: [0, 0]\n'
    assert src.count(old) == 1
    new = '          type: "ContactModel3D"\n          link_name: "flying_arm_3__gripper"\n          position: [0, 0, 0]\n'
    new += '          gains: [%r, %r]\n' % (float(gains[0]), float(gains[1]))
    # (name2 = "zz_elbow" sorts AFTER "end_effector": the gripper's rows come first then)
    new += '        - name: "%s"\n          type: "%s"\n          link_name: "%s"\n          position: [0.1, -0.05, 0.2]\n' % (name2, second, link2)
    if second == "ContactModel6D":
        new += '          orientation: [0, 0, 0, 1]\n'
    new += '          gains: [%r, %r]\n' % (float(gains2[0]), float(gains2[1]))
    src = src.replace(old, new)
    if cone_on_second:  # the friction cone reads the force of the contact on ITS frame (crocoddyl looks the contact up by frame id)
        old_c = '          mu: 0.7\n          link_name: "flying_arm_3__gripper"\n'
        assert src.count(old_c) == 1
        src = src.replace(old_c, '          mu: 0.7\n          link_name: "%s"\n' % link2)
    if extra_frame_cost:  # a frame cost on a THIRD link in the grasp stage: three distinct frames (two contacts + this one)
        old_k = '      contacts:\n        - name: "end_effector"\n'
        assert src.count(old_k) == 1
        src = src.replace(old_k, '        - name: "zz_translation_base"\n          type: "CostModelFrameTranslation"\n          weight: 20\n'
                                 '          link_name: "%s"\n          position: [0, 0, 1.3]\n\n' % extra_frame_cost + old_k)
    if bent is not None:
        # The file's initial state hangs the arm straight down: ANY two points of a stretched chain sit on one line, their constraint
        # rows along that line coincide and Jc M^-1 Jc^T is singular (rank 5) -- and the solver's initial guess repeats that state on
        # every knot.  `bent` = initial joint angles of a regular configuration, for tests that SOLVE the problem.
        old_i = "  initial_state: [-5, 0, 1.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]\n"
        assert src.count(old_i) == 1
        src = src.replace(old_i, "  initial_state: [-5, 0, 1.0, 0, 0, 0, 1, %r, %r, %r, 0, 0, 0, 0, 0, 0, 0, 0, 0]\n" % tuple(float(a) for a in bent))
    f = tmp_path / ("eagle_catch_two_%s_%s_%g_%g_%g_%g_%d%s.yaml" % (second, link2[-6:], gains[0], gains[1], gains2[0], gains2[1], int(cone_on_second),
                                                                  ("" if bent is None else "_bent") + ("" if name2 == "elbow" else "_" + name2) + ("_x" if extra_frame_cost else "")))
    f.write_text(src)
    tr = empc.Trajectory()
    tr.autoSetup(str(f))
    return tr, tr.createProblem(dt, squash, integrator)


def unweighted_barrier_variant(empc, tmp_path, dt=80):
    """displacement with every `limits_state` cost switched from ActivationModelWeightedQuadraticBarrier to the unweighted
    ActivationModelQuadraticBarrier (src/factory/activation.cpp:53-68: bounds only, `weights` ignored) and bounds on every
    component.  No shipped YAML uses the unweighted barrier.  Returns (trajectory, problem)."""
    src = open(empc.yaml_path(CONFIGS["displacement"][0])).read()
    old = '          activation: "ActivationModelWeightedQuadraticBarrier"\n          weights: [0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1]\n'
    new = '          activation: "ActivationModelQuadraticBarrier"\n'
    assert src.count(old) >= 4
    src = src.replace(old, new)
    # an unweighted barrier with zero bounds on the base components would pin the base at the origin: open those bounds
    src = src.replace("u_bound: [0, 0, 0, 0, 0, 0, 1.5, 1.5, 1.5, 0, 0, 0, 0, 0, 0, 3, 3, 3]",
                      "u_bound: [9, 9, 9, 3, 3, 3, 1.5, 1.5, 1.5, 7, 7, 7, 5, 5, 5, 3, 3, 3]")
    src = src.replace("[0, 0, 0, 0, 0, 0, -1.5, -1.5, -1.5, 0, 0, 0, 0, 0, 0, -3, -3, -3]",
                      "[-9, -9, -9, -3, -3, -3, -1.5, -1.5, -1.5, -7, -7, -7, -5, -5, -5, -3, -3, -3]")
    f = tmp_path / "displacement_unweighted_barrier.yaml"
    f.write_text(src)
    tr = empc.Trajectory()
    tr.autoSetup(str(f))
    return tr, tr.createProblem(dt, True, "IntegratedActionModelEuler")


ARM5_CONTACT_STAGE = '''
    - name: "push"
      duration: 520 #ms
      transition: false
      costs:
        - name: "reg_state"
          type: "CostModelState"
          weight: 1e-2
          reference: [0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
          activation: "ActivationModelWeightedQuad"
          weights: [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]

        - name: "reg_control"
          type: "CostModelControl"
          weight: 1e-2
          reference: [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
          activation: "ActivationModelWeightedQuad"
          weights: [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]

        - name: "translation_ee"
          type: "CostModelFrameTranslation"
          weight: 500
          link_name: "flying_arm_5__gripper"
          position: [0.6, 0.1, 0.7]

        - name: "friction_cone"
          type: "CostModelContactFrictionCone"
          weight: 10
          n_surf: [0, 0, 1]
          mu: 0.7
          link_name: "flying_arm_5__gripper"

      contacts:
        - name: "end_effector"
          type: "%(contact)s"
          link_name: "flying_arm_5__gripper"
          position: [0.6, 0.1, 0.7]
%(orientation)s          gains: [%(g0)r, %(g1)r]
'''


def arm5_contact_variant(empc, tmp_path, contact="ContactModel3D", gains=(0.0, 0.0), dt=26):
    """push_slide's robot (hextilt_flying_arm_5: 6 bodies, 6 tilted rotors, 11 velocity dimensions) with a contact stage
    appended -- contact dynamics on the second arm class (src/factory/contacts.cpp:26-79 builds a contact for any robot; no
    shipped YAML has one on this robot).  Returns (trajectory, problem)."""
    src = open(empc.yaml_path(CONFIGS["push_slide"][0])).read()
    src = src.replace("duration: 2000 #ms", "duration: 1040 #ms")
    assert src.count("duration: 1040 #ms") == 1
    src = src.rstrip("\n") + "\n" + ARM5_CONTACT_STAGE % dict(
        contact=contact, orientation='          orientation: [0, 0, 0, 1]\n' if contact == "ContactModel6D" else "",
        g0=float(gains[0]), g1=float(gains[1]))
    f = tmp_path / ("arm5_%s_%g_%g.yaml" % (contact, gains[0], gains[1]))
    f.write_text(src)
    tr = empc.Trajectory()
    tr.autoSetup(str(f))
    return tr, tr.createProblem(dt, True, "IntegratedActionModelEuler")


def arm5_two_contact_variant(empc, tmp_path, gains=(0.0, 0.0), gains2=(0.0, 0.0), dt=26, link2="flying_arm_5__link_3", bent=None):
    """arm5_contact_variant with a SECOND ContactModel3D ("elbow", on `link2`) in the appended stage: two contacts per stage
    (src/stage.cpp:38-48) on the (6, 6) robot class -- 64-lane linearize units (empc_inst_6_6_contact_pair.hip; opt-in)."""
    src = open(empc.yaml_path(CONFIGS["push_slide"][0])).read()
    src = src.replace("duration: 2000 #ms", "duration: 1040 #ms")
    assert src.count("duration: 1040 #ms") == 1
    stage = ARM5_CONTACT_STAGE % dict(contact="ContactModel3D", orientation="", g0=float(gains[0]), g1=float(gains[1]))
    stage = stage.rstrip("\n") + ('\n        - name: "elbow"\n          type: "ContactModel3D"\n          link_name: "%s"\n'
                                  '          position: [0.3, 0.0, 0.9]\n          gains: [%r, %r]\n' % (link2, float(gains2[0]), float(gains2[1])))
    if bent is not None:  # (see two_contact_variant: the stretched arm of the file's initial state is a singular configuration)
        old_i = "  initial_state: [0, 0, 1.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]\n"
        assert src.count(old_i) == 1
        src = src.replace(old_i, "  initial_state: [0, 0, 1.0, 0, 0, 0, 1, %r, %r, %r, %r, %r, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]\n" % tuple(float(a) for a in bent))
    f = tmp_path / ("arm5_two_%g_%g_%g_%g%s.yaml" % (gains[0], gains[1], gains2[0], gains2[1], "" if bent is None else "_bent"))
    f.write_text(src.rstrip("\n") + "\n" + stage)
    tr = empc.Trajectory()
    tr.autoSetup(str(f))
    return tr, tr.createProblem(dt, True, "IntegratedActionModelEuler")


def arm5_mixed_contact_variant(empc, tmp_path, gains6=(5.0, 2.0), dt=26):
    """push_slide's robot with TWO contact stages appended, a ContactModel3D "push" and a ContactModel6D "hold": stages of both
    contact types on the (6, 6) robot class (empc_inst_6_6_contact_mixed.hip; opt-in, EMPC_EXPERIMENTAL_CONTACT=1)."""
    src = open(empc.yaml_path(CONFIGS["push_slide"][0])).read()
    src = src.replace("duration: 2000 #ms", "duration: 780 #ms")
    assert src.count("duration: 780 #ms") == 1
    push = ARM5_CONTACT_STAGE % dict(contact="ContactModel3D", orientation="", g0=0.0, g1=0.0)
    hold = (ARM5_CONTACT_STAGE % dict(contact="ContactModel6D", orientation='          orientation: [0, 0, 0, 1]\n',
                                      g0=float(gains6[0]), g1=float(gains6[1]))).replace('- name: "push"', '- name: "hold"')
    f = tmp_path / ("arm5_mixed_%g_%g.yaml" % tuple(gains6))
    f.write_text(src.rstrip("\n") + "\n" + push.rstrip("\n") + "\n" + hold)
    tr = empc.Trajectory()
    tr.autoSetup(str(f))
    return tr, tr.createProblem(dt, True, "IntegratedActionModelEuler")


SMALL_CLASS_CONTACT_STAGE = '''
    - name: "touch"
      duration: 400 #ms
      transition: false
      costs:
        - name: "reg_state"
          type: "CostModelState"
          weight: 1e-2
          reference: %(xref)s
          activation: "ActivationModelWeightedQuad"
          weights: %(w)s

        - name: "reg_control"
          type: "CostModelControl"
          weight: 1e-2
          reference: %(uref)s
          activation: "ActivationModelWeightedQuad"
          weights: %(uw)s

        - name: "friction_cone"
          type: "CostModelContactFrictionCone"
          weight: 10
          n_surf: [0, 0, 1]
          mu: 0.7
          link_name: "%(link)s"

      contacts:
        - name: "touch"
          type: "%(contact)s"
          link_name: "%(link)s"
          position: [0.1, 0.0, 2.4]
%(orientation)s          gains: [%(g0)r, %(g1)r]
'''

# robot classes without an arm chain long enough for the shipped contact files: (bodies, rotors) = (1, 6), (1, 4), (3, 6)
SMALL_CLASSES = {"hexacopter370": ("hexacopter370/trajectories/hover.yaml", "hexacopter370__base_link"),
                 "iris": ("iris/trajectories/hover.yaml", "iris__base_link"),
                 "hexacopter680_flying_arm_2": ("hexacopter680_flying_arm_2/trajectories/hover.yaml", "hexacopter680__base_link")}


def small_class_contact_variant(empc, tmp_path, robot, contact="ContactModel3D", gains=(0.0, 0.0), dt=40):
    """A hover file of one of the smaller robot classes with a contact stage appended (contact at the base link, friction
    cone cost): src/factory/contacts.cpp:26-79 builds a contact for any robot, no shipped YAML has one on these.  Returns
    (trajectory, problem)."""
    rel, link = SMALL_CLASSES[robot]
    t0 = empc.Trajectory()
    t0.autoSetup(empc.yaml_path(rel))
    xref = [0.0] * t0.nx
    xref[6] = 1.0
    src = open(empc.yaml_path(rel)).read().rstrip("\n") + "\n" + SMALL_CLASS_CONTACT_STAGE % dict(
        xref=xref, w=[1] * t0.ndx, uref=[0] * t0.nu, uw=[1] * t0.nu, link=link, contact=contact,
        orientation='          orientation: [0, 0, 0, 1]\n' if contact == "ContactModel6D" else "", g0=float(gains[0]), g1=float(gains[1]))
    f = tmp_path / ("%s_%s_%g_%g.yaml" % (robot, contact, gains[0], gains[1]))
    f.write_text(src)
    tr = empc.Trajectory()
    tr.autoSetup(str(f))
    return tr, tr.createProblem(dt, True, "IntegratedActionModelEuler")


def mixed_contact_variant(empc, tmp_path, gains6=(0.0, 0.0), dt=32):
    """eagle_catch with its grasp stage followed by a copy of it ("hold") whose contact is a ContactModel6D: a problem with
    stages of BOTH contact types (src/factory/contacts.cpp:26-79 builds either per stage; no shipped YAML mixes them).
    Returns (trajectory, problem)."""
    src = open(empc.yaml_path(CONFIGS["eagle_catch"][0])).read()
    i, j = src.index('    - name: "grasp"'), src.index('    - name: "move_away"')
    grasp = src[i:j]
    old = '          type: "ContactModel3D"\n          link_name: "flying_arm_3__gripper"\n          position: [0, 0, 0]\n          gains: [0, 0]\n'
    assert grasp.count(old) == 1
    hold = grasp.replace('- name: "grasp"', '- name: "hold"').replace(
        old, '          type: "ContactModel6D"\n          link_name: "flying_arm_3__gripper"\n          position: [0, 0, 0]\n'
             '          orientation: [0, 0, 0, 1]\n          gains: [%r, %r]\n' % (float(gains6[0]), float(gains6[1])))
    f = tmp_path / ("eagle_catch_mixed_%g_%g.yaml" % tuple(gains6))
    f.write_text(src[:j] + hold + src[j:])
    tr = empc.Trajectory()
    tr.autoSetup(str(f))
    return tr, tr.createProblem(dt, True, "IntegratedActionModelEuler")
