// GstAllocator for HIP device memory (see mvfxhipmemory.h).
#include "mvfxhipmemory.h"

#include "mi355vfx.h"

#include <string.h>

typedef struct {
    GstMemory mem;
    void *dptr;       // hipMalloc'ed
    guint8 *shadow;   // host copy while CPU-mapped
    GstMapFlags shadow_flags;
    gint cpu_maps;
    GMutex lock;
} MvfxHipMemory;

typedef struct { GstAllocator parent; } MvfxHipAllocator;
typedef struct { GstAllocatorClass parent_class; } MvfxHipAllocatorClass;

G_DEFINE_TYPE(MvfxHipAllocator, mvfx_hip_allocator, GST_TYPE_ALLOCATOR)

// Small free list keyed by size: video buffers of one stream all have the same size, so a freed
// block is handed to the next allocation instead of paying hipFree + hipMalloc per frame.
#define MVFX_FREELIST_MAX 16
static GMutex freelist_lock;
static struct { void *dptr; gsize size; } freelist[MVFX_FREELIST_MAX];

static void *freelist_take(gsize size)
{
    void *p = NULL;
    g_mutex_lock(&freelist_lock);
    for (int i = 0; i < MVFX_FREELIST_MAX; i++)
        if (freelist[i].dptr && freelist[i].size == size) { p = freelist[i].dptr; freelist[i].dptr = NULL; break; }
    g_mutex_unlock(&freelist_lock);
    return p;
}

static gboolean freelist_give(void *dptr, gsize size)
{
    gboolean kept = FALSE;
    g_mutex_lock(&freelist_lock);
    for (int i = 0; i < MVFX_FREELIST_MAX && !kept; i++)
        if (!freelist[i].dptr) { freelist[i].dptr = dptr; freelist[i].size = size; kept = TRUE; }
    g_mutex_unlock(&freelist_lock);
    return kept;
}

static GstMemory *mvfx_hip_alloc(GstAllocator *allocator, gsize size, GstAllocationParams *params)
{
    void *dptr = freelist_take(size);
    if (!dptr && mvfx_device_alloc(&dptr, size) != MVFX_OK) {
        GST_ERROR("HIP allocation of %" G_GSIZE_FORMAT " bytes failed: %s", size, mvfx_last_error());
        return NULL;
    }
    MvfxHipMemory *m = g_new0(MvfxHipMemory, 1);
    gst_memory_init(GST_MEMORY_CAST(m), (GstMemoryFlags)0, allocator, NULL, size, 255, 0, size);
    m->dptr = dptr;
    g_mutex_init(&m->lock);
    return GST_MEMORY_CAST(m);
}

static void mvfx_hip_free(GstAllocator *, GstMemory *mem)
{
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (!freelist_give(m->dptr, mem->maxsize))
        mvfx_device_free(m->dptr);
    g_free(m->shadow);
    g_mutex_clear(&m->lock);
    g_free(m);
}

static gpointer mvfx_hip_map_full(GstMemory *mem, GstMapInfo *info, gsize maxsize)
{
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (info->flags & MVFX_MAP_HIP)
        return m->dptr; // device pointer, zero copy
    g_mutex_lock(&m->lock);
    if (m->cpu_maps == 0) {
        m->shadow = (guint8 *)g_malloc(mem->maxsize);
        m->shadow_flags = (GstMapFlags)0;
        // always fetch: a partial WRITE map must not lose the bytes it does not touch
        if (mvfx_copy_to_host(m->shadow, m->dptr, mem->maxsize, NULL) != MVFX_OK) {
            g_free(m->shadow);
            m->shadow = NULL;
            g_mutex_unlock(&m->lock);
            return NULL;
        }
    }
    m->cpu_maps++;
    m->shadow_flags = (GstMapFlags)(m->shadow_flags | info->flags);
    g_mutex_unlock(&m->lock);
    return m->shadow;
}

static void mvfx_hip_unmap_full(GstMemory *mem, GstMapInfo *info)
{
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (info->flags & MVFX_MAP_HIP)
        return;
    g_mutex_lock(&m->lock);
    if (--m->cpu_maps == 0) {
        if (m->shadow_flags & GST_MAP_WRITE)
            mvfx_copy_to_device(m->dptr, m->shadow, mem->maxsize, NULL);
        g_free(m->shadow);
        m->shadow = NULL;
    }
    g_mutex_unlock(&m->lock);
}

static GstMemory *mvfx_hip_copy(GstMemory *mem, gssize offset, gssize size)
{
    MvfxHipMemory *m = (MvfxHipMemory *)mem;
    if (size == -1)
        size = mem->size > (gsize)offset ? mem->size - offset : 0;
    GstMemory *copy = mvfx_hip_alloc(mem->allocator, size, NULL);
    if (!copy)
        return NULL;
    if (mvfx_copy_device_to_device(((MvfxHipMemory *)copy)->dptr, (guint8 *)m->dptr + mem->offset + offset, size, NULL) != MVFX_OK) {
        gst_memory_unref(copy);
        return NULL;
    }
    return copy;
}

static void mvfx_hip_allocator_class_init(MvfxHipAllocatorClass *klass)
{
    GstAllocatorClass *ac = GST_ALLOCATOR_CLASS(klass);
    ac->alloc = mvfx_hip_alloc;
    ac->free = mvfx_hip_free;
}

static void mvfx_hip_allocator_init(MvfxHipAllocator *self)
{
    GstAllocator *a = GST_ALLOCATOR_CAST(self);
    a->mem_type = MVFX_HIP_MEMORY_TYPE;
    a->mem_map_full = mvfx_hip_map_full;
    a->mem_unmap_full = mvfx_hip_unmap_full;
    a->mem_copy = mvfx_hip_copy;
    // no mem_share: sub-buffers of device memory are not needed by these elements
    GST_OBJECT_FLAG_SET(a, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

GstAllocator *mvfx_hip_allocator_get(void)
{
    static gsize once = 0;
    static GstAllocator *singleton = NULL;
    if (g_once_init_enter(&once)) {
        singleton = (GstAllocator *)g_object_new(mvfx_hip_allocator_get_type(), NULL);
        gst_object_ref_sink(singleton);
        g_once_init_leave(&once, 1);
    }
    return (GstAllocator *)gst_object_ref(singleton);
}

gboolean mvfx_is_hip_memory(GstMemory *mem)
{
    return mem && mem->allocator && g_strcmp0(mem->allocator->mem_type, MVFX_HIP_MEMORY_TYPE) == 0;
}

gboolean mvfx_buffer_is_hip(GstBuffer *buf)
{
    return buf && gst_buffer_n_memory(buf) == 1 && mvfx_is_hip_memory(gst_buffer_peek_memory(buf, 0));
}

gboolean mvfx_caps_has_hip_feature(const GstCaps *caps)
{
    if (!caps || gst_caps_is_empty(caps) || gst_caps_is_any(caps))
        return FALSE;
    GstCapsFeatures *f = gst_caps_get_features(caps, 0);
    return f && gst_caps_features_contains(f, MVFX_CAPS_FEATURE_MEMORY_HIP);
}

GstCaps *mvfx_caps_set_hip_feature(const GstCaps *caps, gboolean hip)
{
    GstCaps *out = gst_caps_copy(caps);
    for (guint i = 0; i < gst_caps_get_size(out); i++)
        gst_caps_set_features(out, i, hip ? gst_caps_features_new(MVFX_CAPS_FEATURE_MEMORY_HIP, NULL)
                                          : gst_caps_features_new_empty());
    return out;
}

GstCaps *mvfx_caps_with_hip_feature(const GstCaps *system_caps)
{
    return mvfx_caps_set_hip_feature(system_caps, TRUE);
}
