/*
 * activation.c -- ActivationFunction handles (reference: layers/activation.{h,c},
 * layers/activation_default.{h,c}).  The five built-ins are tagged with a kind the
 * device kernels understand; handles made with ActivationFunctionCreate carry the
 * caller's host callbacks and are tagged CUSTOM.
 */
#include <stdlib.h>
#include "nntk_internal.h"

static ActivationFunction make(int kind, int size, float relu_a, int vector_size) {
    ActivationFunction f = (ActivationFunction)calloc(1, sizeof(struct ActivationFunctionStruct));
    if (!f) return NULL;
    f->kind = kind;
    f->input_size = size;
    f->relu_a = relu_a;
    f->vector_size = vector_size;
    return f;
}

/* activation.c:34-45 */
ActivationFunction ActivationFunctionCreate(int size, ActivationImplementerDestroy destroy_fn, void *implementer,
                                            ActivationFunctionImpl function, ActivationFunctionDerivative derivative,
                                            ActivationFunctionDerivative cached_derivative) {
    ActivationFunction f = make(NNTK_ACT_CUSTOM, size, 1.0f, 0);
    if (!f) return NULL;
    f->implementer = implementer;
    f->destroy_fn = destroy_fn;
    f->function = function;
    f->derivative = derivative;
    f->cached_derivative = cached_derivative;
    return f;
}

/* activation.c:27-32 */
void ActivationFunctionDestroy(ActivationFunction filter) {
    if (!filter) return;
    if (filter->implementer && filter->destroy_fn) filter->destroy_fn(filter->implementer);
    free(filter);
}

/* activation_default.c:105, :176, :49, :135, :81 */
ActivationFunction ActivationFunctionCreateIdentity(int input_size) { return make(NNTK_ACT_IDENTITY, input_size, 1.0f, 0); }
ActivationFunction ActivationFunctionCreateSoftmax(int input_size, int vector_size) {
    return make(NNTK_ACT_SOFTMAX, input_size, 1.0f, vector_size);
}
ActivationFunction ActivationFunctionCreateSigmoid(int input_size) { return make(NNTK_ACT_SIGMOID, input_size, 1.0f, 0); }
ActivationFunction ActivationFunctionCreateReLU(int input_size, float a) { return make(NNTK_ACT_RELU, input_size, a, 0); }
ActivationFunction ActivationFunctionCreateTanh(int input_size) { return make(NNTK_ACT_TANH, input_size, 1.0f, 0); }

int nntk_act_kind(ActivationFunction a) { return a ? a->kind : NNTK_ACT_NONE; }

int nntk_act_fusable(ActivationFunction a) {
    if (!a) return 1;
    return a->kind == NNTK_ACT_IDENTITY || a->kind == NNTK_ACT_SIGMOID || a->kind == NNTK_ACT_TANH ||
           a->kind == NNTK_ACT_RELU;
}

static long act_elems(ActivationFunction f, int size) {
    long n = size > 0 ? size : f->input_size;
    return f->kind == NNTK_ACT_SOFTMAX ? n * (long)f->vector_size : n;
}

int ActivationFunctionApplyDevice(ActivationFunction filter, const float *d_input, float *d_output, int size) {
    nntk_shim_clear_error();
    if (!filter) NNTK_FAIL("ActivationFunctionApplyDevice: NULL handle");
    if (filter->kind == NNTK_ACT_CUSTOM)
        NNTK_FAIL("custom host-callback activation cannot run on device pointers");
    return nntk_shim_activation(filter->kind, filter->relu_a, filter->vector_size, d_input, d_output,
                                act_elems(filter, size));
}

/* activation.c:23-25.  Built-ins run on the GPU (upload, kernel, download); a
 * custom handle is the caller's own host function and is simply invoked. */
void ActivationFunctionApply(ActivationFunction filter, const float *input, float *output) {
    nntk_shim_clear_error();
    if (!filter) { nntk_set_error("ActivationFunctionApply: NULL handle"); return; }
    if (filter->kind == NNTK_ACT_CUSTOM) {
        if (filter->function) filter->function(filter->implementer, input, output, filter->input_size);
        return;
    }
    long n = act_elems(filter, 0);
    float *d = (float *)nntk_shim_malloc((size_t)n * sizeof(float));
    if (!d) return;
    if (nntk_shim_upload(d, input, (size_t)n * sizeof(float)) == 0 &&
        nntk_shim_activation(filter->kind, filter->relu_a, filter->vector_size, d, d, n) == 0)
        nntk_shim_download(output, d, (size_t)n * sizeof(float));
    nntk_shim_free(d);
}
