"""Scene compiler of the design API -- own implementation, file-format compatible with the reference.

A design script builds a tree of components (brush + homogeneous transform, optionally subtractive)
under a root scaled by 5, and `commit()` writes the compiled scene into the working directory:

    scene.cl            OpenCL C: AD offsets, user defines, helper functions, sd<N>/shader<N>, the banks
    scene.txt           one line per component in pre-order: brush material pos right up forward
    buildprocedure.txt  the register program "opcode left right dest" that combines the components
    arbitrary_data.hex  131072 little-endian floats

These files are the contract with the native side (include/dcsg.h: dcsg_load_scene); their formats
follow the reference writer (scenecompiler.py:466-582 of mikeandike523/DesignCSG) byte for byte, which
tests/test_frontend.py pins against fixtures captured from the reference itself.  The public names
(Transform, Component, IntersectionComponent, SceneCompiler, ...) match the reference module so
that existing design scripts run unchanged.
"""
import enum

import numpy as np

INITIAL_SCALE = 5.0
ARBITRARY_DATA_POINTS = 131072

# opcode numbers shared with the evaluator (k2.cl:12-17)
COMMAND_VALUES = {"IMPORT": 0, "EXPORT": 1, "MIN": 2, "MAX": 3, "NEGATE": 4, "IDENTITY": 5,
                  # extension opcodes of this package (not in the reference; smooth joins)
                  "SMOOTH_MIN": 6, "SMOOTH_MAX": 7, "SET_K": 8}


def matmul(*mats):
    """Product of two or more matrices, associated from the right: A @ (B @ (C @ ...))."""
    if len(mats) < 2:
        raise TypeError("Insufficient Arguments")
    acc = np.matmul(mats[-2], mats[-1])
    for m in reversed(mats[:-2]):
        acc = np.matmul(m, acc)
    return acc


class Utils:
    @staticmethod
    def fwrite(fname, content):
        with open(fname, "w") as fl:
            fl.write(content)


def _h4(rows):
    return np.array(rows, dtype=float)


class Transform:
    """4x4 homogeneous transforms, column-vector convention (p' = M p)."""

    @staticmethod
    def homogenize(v):
        return np.array([v[0], v[1], v[2], 0], dtype=float)

    @staticmethod
    def axes(v1, v2, v3):
        # the three vectors become the COLUMNS of the linear part
        return _h4([Transform.homogenize(v1), Transform.homogenize(v2), Transform.homogenize(v3), [0, 0, 0, 1]]).T

    @staticmethod
    def translation(offset):
        m = np.identity(4)
        m[0, 3], m[1, 3], m[2, 3] = offset[0], offset[1], offset[2]
        return m

    @staticmethod
    def to_homogenous(v):
        return np.concatenate((v, [1.0]))

    @staticmethod
    def from_homogenous(v):
        return v[0:3]

    @staticmethod
    def reciprocal_vector(v):
        return v / (np.linalg.norm(v) ** 2)

    @staticmethod
    def eulerY(yaw):
        a = -yaw
        return _h4([[np.cos(a), 0, np.sin(a), 0],
                    [0, 1, 0, 0],
                    [np.cos(a + np.pi / 2), 0, np.sin(a + np.pi / 2), 0],
                    [0, 0, 0, 1]]).T

    @staticmethod
    def eulerX(pitch):
        return _h4([[1, 0, 0, 0],
                    [0, np.sin(pitch + np.pi / 2), np.cos(pitch + np.pi / 2), 0],
                    [0, np.sin(pitch), np.cos(pitch), 0],
                    [0, 0, 0, 1]]).T

    @staticmethod
    def eulerZ(roll):
        return _h4([[np.cos(roll), np.sin(roll), 0, 0],
                    [np.cos(roll + np.pi / 2.0), np.sin(roll + np.pi / 2.0), 0, 0],
                    [0, 0, 1, 0],
                    [0, 0, 0, 1]]).T

    @staticmethod
    def scaling(scale):
        return _h4([[scale[0], 0, 0, 0], [0, scale[1], 0, 0], [0, 0, scale[2], 0], [0, 0, 0, 1]]).T

    @staticmethod
    def rotation(yaw, pitch, roll):
        return matmul(Transform.eulerY(yaw), Transform.eulerX(pitch), Transform.eulerZ(roll))

    @staticmethod
    def initial(position, yaw, pitch, roll, scale):
        return matmul(Transform.translation(position), Transform.rotation(yaw, pitch, roll), Transform.scaling(scale))

    @staticmethod
    def normalized(v):
        return v / np.linalg.norm(v)

    @staticmethod
    def identity():
        return Transform.axes([1, 0, 0], [0, 1, 0], [0, 0, 1])


class ArgumentType(enum.Enum):
    IMMEDIATE = enum.auto()
    ALLOCATION = enum.auto()


class Argument:
    """Operand of a build command: a stack slot (ALLOCATION) or a literal (IMMEDIATE, -1 = none)."""

    def __init__(self, type, address):
        self.type = type
        self.address = address

    @staticmethod
    def null():
        return Argument(ArgumentType.IMMEDIATE, -1)

    @staticmethod
    def immediate(v):
        return Argument(ArgumentType.IMMEDIATE, v)

    def __eq__(self, other):
        return isinstance(other, Argument) and (self.type, self.address) == (other.type, other.address)

    def __repr__(self):
        return "Argument(type=%r, address=%r)" % (self.type, self.address)


class Command:
    def __init__(self, command_code, left_argument, right_argument, destination):
        self.command_code = command_code
        self.left_argument = left_argument
        self.right_argument = right_argument
        self.destination = destination

    def _operands(self):
        return (self.left_argument.address, self.right_argument.address, self.destination.address)

    def __repr__(self):
        return "%s %d %d %d" % ((self.command_code,) + self._operands())

    def __str__(self):
        return "%d %d %d %d" % ((COMMAND_VALUES[self.command_code],) + self._operands())


class Incrementor:
    def __init__(self):
        self._count = 0

    def count(self):
        return self._count

    def preincremented(self):
        self._count += 1
        return self._count

    def postincremented(self):
        self._count += 1
        return self._count - 1


class Allocator:
    """Hands out interpreter stack slots in order; named slots become attributes (e.g. R0)."""

    def __init__(self):
        self.next_free_address = Incrementor()
        self.allocations = {}

    def allocate(self, **kwargs):
        arg = Argument(ArgumentType.ALLOCATION, self.next_free_address.postincremented())
        name = kwargs.get("name", "ALLOC_%d" % arg.address)
        self.allocations[name] = arg
        setattr(self, name, arg)
        return arg


class Brush:
    """OpenCL C body of one signed-distance function `float sd<N>(float3 v)`."""

    def __init__(self, **kwargs):
        self.body = kwargs["body"]
        self.bank_index = kwargs["bank_index"]

    def __str__(self):
        return "\n        float sd%d( float3 v){\n\n            %s\n\n        }\n        " % (self.bank_index, self.body)


class Material:
    def __init__(self, **kwargs):
        self.body = kwargs["body"]
        self.bank_index = kwargs["bank_index"]

    def __str__(self):
        return ("\n        float3 shader%d (float3 gv, float3 lv, float3 n){\n\n            %s\n\n        }\n        "
                % (self.bank_index, self.body))


class Component:
    """Node of the instance tree: a brush under a transform, with children joined by union (or by
    intersection for IntersectionComponent); `subtractive` nodes are carved out of their parent."""

    join_mode = "MIN"

    def __init__(self, **kwargs):
        self.brush = kwargs["brush"]
        self.material = kwargs["material"]
        self.intrinsic_transform = kwargs.get("transform", np.identity(4))
        self.subtractive = kwargs.get("subtractive", False)
        self.inherited_transform = np.identity(4)
        self.propogated_transform = np.identity(4)
        self.children = []
        self.parent = None

    def add_child(self, child):
        child.parent = self
        self.children.append(child)

    def fabricate(self, **kwargs):
        """Instance of this prefab under `transform`; children are re-instanced unchanged."""
        inst = Component(brush=self.brush, material=self.material,
                         transform=matmul(kwargs["transform"], self.intrinsic_transform),
                         subtractive=kwargs.get("subtractive", self.subtractive))
        for child in self.children:
            inst.add_child(child=child.fabricate(transform=np.identity(4)))
        return inst

    def propogate_transforms(self):
        m = self.intrinsic_transform
        node = self.parent
        while node is not None:
            m = matmul(node.intrinsic_transform, m)
            node = node.parent
        self.propogated_transform = m

    def apply_transform(self, tf):
        self.intrinsic_transform = matmul(tf, self.intrinsic_transform)

    def position(self):
        return Transform.from_homogenous(matmul(self.propogated_transform, Transform.to_homogenous(np.zeros(3))))

    def _axis(self, i):
        return self.propogated_transform.T[i, 0:3].squeeze()

    def right(self):
        return self._axis(0)

    def up(self):
        return self._axis(1)

    def forward(self):
        return self._axis(2)

    def get_unrolled_components(self):
        out = [self]
        for child in self.children:
            out.extend(child.get_unrolled_components())
        return out

    def get_commands(self, allocator, joinMode=None):
        """Register program of this subtree.  A node with children imports its own brush into its slot,
        then joins every child: leaves through the scratch register R0, groups through their own slot."""
        join = joinMode or "MIN"
        if not self.children:
            return []
        me, r0 = self.variable, allocator.R0
        blend = getattr(self, "blend", None)     # extension: smooth groups (SmoothComponent)
        if blend is not None:
            join = "SMOOTH_" + join
            bits = int(np.array([blend], dtype="<f4").view("<i4")[0])
            set_k = Command("SET_K", Argument.immediate(bits), Argument.immediate(0), Argument.immediate(0))
        cmds = [Command("IMPORT", Argument.immediate(self.brush.bank_index), Argument.immediate(self.unrolled_index), me)]
        for child in self.children:
            if child.children:
                cmds.extend(child.get_commands(allocator))
                source = child.variable
            else:
                cmds.append(Command("IMPORT", Argument.immediate(child.brush.bank_index),
                                    Argument.immediate(child.unrolled_index), r0))
                source = r0
            if child.subtractive:
                cmds.append(Command("NEGATE", source, Argument.null(), r0))
                if blend is not None:
                    cmds.extend([set_k, Command("SMOOTH_MAX", me, r0, me)])
                else:
                    cmds.append(Command("MAX", me, r0, me))
            else:
                if blend is not None:
                    cmds.append(set_k)       # right before the join: a nested smooth group may have changed it
                cmds.append(Command(join, me, source, me))
        return cmds


class _IntersectionComponent(Component):
    """Group whose children are intersected: starts from the void brush (distance 0) and joins with MAX."""

    def __init__(self, compiler, **kwargs):
        kwargs = dict(kwargs)
        kwargs["brush"] = compiler.void_brush()
        super().__init__(**kwargs)

    def get_commands(self, allocator):
        return super().get_commands(allocator, "MAX")


class _SmoothComponent(Component):
    """EXTENSION (not in the reference): group whose children are joined with the polynomial smooth minimum
    (smooth union) of blend radius `blend`; with intersection=True it starts from the void brush and joins with
    the smooth maximum.  The group itself joins its parent like any other group."""

    def __init__(self, compiler, blend, intersection=False, **kwargs):
        kwargs = dict(kwargs)
        if not (np.isfinite(blend) and blend > 0):
            raise ValueError("blend radius must be finite and > 0")
        if intersection:
            kwargs["brush"] = compiler.void_brush()
        super().__init__(**kwargs)
        self.blend = float(np.float32(blend))
        self.intersection = intersection

    def get_commands(self, allocator):
        return super().get_commands(allocator, "MAX" if self.intersection else "MIN")


class ArbitraryDataChunk:
    def __init__(self, name, start, data):
        self.name = name
        self.start = start
        self.data = data


_BASIC_LIGHTING = """
        
        float3 n_g = n.x*rgt_g+n.y*upp_g+n.z*fwd_g;

        float L = dot(n_g,(float3)(0.0,0.0,-1.0)); return (float3)(L,L,L);



        """


class _SceneCompiler:
    """Process-wide scene under construction (singleton, see SceneCompiler())."""

    def __init__(self, **kwargs):
        self.adCounter = 0
        self.ad = []
        self.brush_counter = Incrementor()
        self.material_counter = Incrementor()
        self.brushes = []
        self.materials = []
        self.empty_brush = self.define_brush(body="return MAX_DISTANCE;")   # brush 0: nothing
        self.space_brush = self.define_brush(body="return 0.0;")            # brush 1: everything
        self.abs_normals = self.define_material(body="return fabs(n);")
        self.basic_lighting = self.define_material(body=_BASIC_LIGHTING)
        self.root = Component(brush=self.null_brush(), material=self.default_material(),
                              transform=Transform.scaling(np.array([INITIAL_SCALE] * 3)))
        self.allocator = Allocator()
        self.auxillary_functions = [" "]
        self.preprocessor_defines = []

    def define_auxillary_function(self, function):
        self.auxillary_functions.append(function)

    def add_preprocessor_define(self, define):
        self.preprocessor_defines.append(define)

    def define_brush(self, **kwargs):
        self.brushes.append(Brush(body=kwargs["body"], bank_index=self.brush_counter.postincremented()))
        return self.brushes[-1]

    def define_material(self, **kwargs):
        self.materials.append(Material(body=kwargs["body"], bank_index=self.material_counter.postincremented()))
        return self.materials[-1]

    def null_brush(self):
        return self.empty_brush

    def void_brush(self):
        return self.space_brush

    def default_material(self):
        return self.basic_lighting

    def addArbitraryData(self, name, data):
        self.ad.append(ArbitraryDataChunk(name, self.adCounter, data))
        self.adCounter += len(data)

    # ---- writers ------------------------------------------------------------------------------
    def _scene_cl(self):
        ind = " " * 8
        ad_defs = "".join("#define AD_%s %d\n" % (c.name, c.start) for c in self.ad)
        sdf_cases = "\n".join("\ncase %d: return sd%d(v); break;\n" % (b.bank_index, b.bank_index) for b in self.brushes)
        shader_cases = "\n".join("\ncase %d: return shader%d(gv,lv,n); break;\n" % (m.bank_index, m.bank_index)
                                 for m in self.materials)
        parts = [
            "\n" + ind + "\n", ad_defs, "\n\n",
            ind, "\n".join(self.preprocessor_defines), "\n\n",
            ind, "\n".join(self.auxillary_functions), "\n\n",
            ind, "\n".join(str(b) for b in self.brushes), "\n\n",
            ind, "\n".join(str(m) for m in self.materials), "\n\n\n",
            ind, "float sdf_bank(float3 v, unsigned char shape_id){\n\n",
            ind, "    switch(shape_id){\n\n",
            ind, "        ", sdf_cases, "\n\n",
            ind, "    }\n\n",
            ind, "    return 0.0;\n\n",
            ind, "}\n\n",
            ind, "float3 shader_bank(float3 gv, float3 lv, float3 n, unsigned char material_id){\n\n\n",
            ind, "    switch(material_id){\n\n",
            ind, "        ", shader_cases, "\n\n",
            ind, "    }\n\n",
            ind, "    return (float3)(1.0, 1.0, 1.0);\n",
            ind, "}\n",
            ind, "\n", ind, "\n", ind,
        ]
        return "".join(parts)

    def commit(self):
        Utils.fwrite("scene.cl", self._scene_cl())

        components = self.root.get_unrolled_components()
        for index, comp in enumerate(components):
            comp.unrolled_index = index
            comp.propogate_transforms()

        rows = []
        for comp in components:
            vals = list(comp.position())
            for axis in (comp.right(), comp.up(), comp.forward()):
                vals.extend(Transform.reciprocal_vector(axis))
            rows.append("%d %d " % (comp.brush.bank_index, comp.material.bank_index)
                        + " ".join("{:.6f}".format(v) for v in vals) + "\n")
        Utils.fwrite("scene.txt", "".join(rows))

        # one stack slot per component that has children, in pre-order, then the scratch register
        for comp in components:
            if comp.children:
                comp.variable = self.allocator.allocate()
        export_variable = self.root.variable
        self.allocator.allocate(name="R0")
        commands = self.root.get_commands(self.allocator)
        commands.append(Command("EXPORT", export_variable, Argument.null(), Argument.null()))
        Utils.fwrite("buildprocedure.txt", "\n".join(str(c) for c in commands))

        table = np.zeros(ARBITRARY_DATA_POINTS, dtype="<f4")
        for chunk in self.ad:
            table[chunk.start:chunk.start + len(chunk.data)] = np.asarray(chunk.data, dtype="<f4")
        with open("arbitrary_data.hex", "wb") as fl:
            fl.write(table.tobytes())

        print("Instance tree compiled successfully.")


compiler = _SceneCompiler()


def SceneCompiler():
    return compiler


def IntersectionComponent(**kwargs):
    return _IntersectionComponent(compiler, **kwargs)


def SmoothComponent(blend, intersection=False, **kwargs):
    return _SmoothComponent(compiler, blend, intersection, **kwargs)
