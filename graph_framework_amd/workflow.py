"""Host-side mirror of workflow::manager (graph_framework/workflow.hpp:215-425).

Same names and call order as the reference: add_preitem / add_item /
add_converge_item, compile, pre_run, run, wait, copy_to_device, copy_to_host,
check_value.  Work items arrive as GFIR (the DAG the reference front end built,
serialized by gfir_serialize.hpp) instead of node lists; variables are named
by their keys.
"""
import numpy as np

from .backend import Context, key_of


class WorkItem:
    """workflow::work_item (workflow.hpp:22-76)."""

    def __init__(self, context, gfir, inputs, outputs, size, initial=None):
        self.kernel = context.add_kernel(gfir, size)
        self.inputs = list(inputs)
        self.outputs = list(outputs)
        self.initial = initial
        self.size = size

    def create_kernel_call(self):
        init = None
        if self.initial is not None:
            init = [self.initial.get(name) for name in self.inputs]
        self.kernel.create_kernel_call(self.inputs, self.outputs, init)

    def run(self, steps=1):
        self.kernel.run(steps)


class ConvergeItem(WorkItem):
    """workflow::converge_item (workflow.hpp:129-205): rerun until the max of the last
    output stalls."""

    def __init__(self, context, gfir, inputs, outputs, size, initial=None,
                 tolerance=1.0e-30, max_iterations=1000):
        super().__init__(context, gfir, inputs, outputs, size, initial)
        self.tolerance = tolerance
        self.max_iterations = max_iterations
        self.iterations = None
        self.last_max = None

    def run(self, steps=1):
        self.iterations, self.last_max = self.kernel.converge(self.tolerance, self.max_iterations)


class Manager:
    """workflow::manager<T, SAFE_MATH>(index) (workflow.hpp:215-425)."""

    def __init__(self, index=0, stream=None):
        self.context = Context(index, stream)
        self.preitems = []
        self.items = []

    def add_preitem(self, gfir, inputs, outputs, size, initial=None):
        item = WorkItem(self.context, gfir, inputs, outputs, size, initial)
        self.preitems.append(item)
        return item

    def add_item(self, gfir, inputs, outputs, size, initial=None):
        item = WorkItem(self.context, gfir, inputs, outputs, size, initial)
        self.items.append(item)
        return item

    def add_converge_item(self, gfir, inputs, outputs, size, initial=None,
                          tolerance=1.0e-30, max_iterations=1000):
        item = ConvergeItem(self.context, gfir, inputs, outputs, size, initial, tolerance, max_iterations)
        self.items.append(item)
        return item

    def compile(self):
        """manager::compile (workflow.hpp:336-345): build the module, then bind every item."""
        self.context.compile()
        for item in self.preitems + self.items:
            item.create_kernel_call()

    def pre_run(self):
        for item in self.preitems:
            item.run()

    def run(self, steps=1):
        for item in self.items:
            item.run(steps)

    def wait(self):
        self.context.wait()

    def copy_to_device(self, name, source):
        self.context.copy_to_device(name, source)

    def copy_to_host(self, name, destination):
        return self.context.copy_to_host(name, destination)

    def check_value(self, index, name):
        return self.context.check_value(index, name)

    def get_context(self):
        return self.context
