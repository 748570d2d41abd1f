/*
 * streamkit_native_abi.h — StreamKit native-plugin C ABI, version 2, as the HOST CALLS IT.
 *
 * Restated from the reference's Rust definition (the in-tree C header
 * examples/plugins/gain-native-c/streamkit_plugin.h lags it: it still shows
 * 5-argument process_packet / 3-argument flush):
 *   /root/reference/sdks/plugin-sdk/native/src/types.rs:13-264   (every struct below, field for field)
 *   /root/reference/crates/plugin-native/src/lib.rs:50-103       (load: symbol lookup, version check)
 *   /root/reference/crates/plugin-native/src/wrapper.rs:159-191  (create_instance)
 *   /root/reference/crates/plugin-native/src/wrapper.rs:424-432  (process_packet: 7 arguments)
 *   /root/reference/crates/plugin-native/src/wrapper.rs:346-352  (flush: 5 arguments)
 *
 * All Rust `#[repr(C)]` enums are C `int`.  Everything that crosses is borrowed for
 * the duration of the call (conversions.rs:213-304, 322-399); error strings are
 * borrowed until the next error on the same OS thread (conversions.rs:441-461).
 */
#ifndef STREAMKIT_NATIVE_ABI_H
#define STREAMKIT_NATIVE_ABI_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define STREAMKIT_NATIVE_PLUGIN_API_VERSION 2u            /* types.rs:13 */
#define STREAMKIT_PLUGIN_API_SYMBOL "streamkit_native_plugin_api" /* types.rs:264 */

typedef void* CPluginHandle;                               /* types.rs:16 */

typedef enum { SK_LOG_TRACE = 0, SK_LOG_DEBUG = 1, SK_LOG_INFO = 2, SK_LOG_WARN = 3, SK_LOG_ERROR = 4 } CLogLevel; /* types.rs:21-27 */
typedef void (*CLogCallback)(CLogLevel level, const char* target, const char* message, void* user_data);            /* types.rs:35 */

typedef struct { bool success; const char* error_message; } CResult;                                                /* types.rs:40-49 */

typedef enum { SK_SAMPLE_F32 = 0, SK_SAMPLE_S16LE = 1 } CSampleFormat;                                               /* types.rs:64-67 */
typedef struct { uint32_t sample_rate; uint16_t channels; CSampleFormat sample_format; } CAudioFormat;              /* types.rs:72-76 */

typedef enum {                                                                                                       /* types.rs:81-90 */
    SK_PACKET_RAW_AUDIO = 0, SK_PACKET_OPUS_AUDIO = 1, SK_PACKET_TEXT = 2, SK_PACKET_TRANSCRIPTION = 3,
    SK_PACKET_CUSTOM = 4, SK_PACKET_BINARY = 5, SK_PACKET_ANY = 6, SK_PACKET_PASSTHROUGH = 7
} CPacketType;
typedef enum { SK_CUSTOM_JSON = 0 } CCustomEncoding;                                                                 /* types.rs:95-97 */

typedef struct {                                                                                                     /* types.rs:102-109 */
    uint64_t timestamp_us; bool has_timestamp_us;
    uint64_t duration_us;  bool has_duration_us;
    uint64_t sequence;     bool has_sequence;
} CPacketMetadata;

typedef struct {                                                                                                     /* types.rs:115-122 */
    const char* type_id; CCustomEncoding encoding; const uint8_t* data_json; size_t data_len; const CPacketMetadata* metadata;
} CCustomPacket;

typedef struct { CPacketType type_discriminant; const CAudioFormat* audio_format; const char* custom_type_id; } CPacketTypeInfo; /* types.rs:128-134 */

typedef struct { uint32_t sample_rate; uint16_t channels; const float* samples; size_t sample_count; } CAudioFrame; /* types.rs:138-143 */

/* RawAudio: data -> CAudioFrame, len = sizeof; Text: NUL-terminated, len includes NUL;
 * Transcription: data = UTF-8 JSON bytes of TranscriptionData, len = byte count (conversions.rs:249-260, 356-361) */
typedef struct { CPacketType packet_type; const void* data; size_t len; } CPacket;                                  /* types.rs:148-152 */

typedef struct { const char* name; const CPacketTypeInfo* accepts_types; size_t accepts_types_count; } CInputPin;   /* types.rs:156-161 */
typedef struct { const char* name; CPacketTypeInfo produces_type; } COutputPin;                                     /* types.rs:165-168 */

typedef struct {                                                                                                     /* types.rs:172-185 */
    const char* kind; const char* description;
    const CInputPin* inputs; size_t inputs_count;
    const COutputPin* outputs; size_t outputs_count;
    const char* param_schema;
    const char* const* categories; size_t categories_count;
} CNodeMetadata;

typedef CResult (*COutputCallback)(const char* pin_name, const CPacket* packet, void* user_data);                   /* types.rs:189 */
/* Option<extern "C" fn>: may be NULL */
typedef CResult (*CTelemetryCallback)(const char* event_type, const uint8_t* data_json, size_t data_len,
                                      const CPacketMetadata* metadata, void* user_data);                            /* types.rs:199-201 */

typedef struct {                                                                                                     /* types.rs:206-261 */
    uint32_t version;
    const CNodeMetadata* (*get_metadata)(void);
    CPluginHandle (*create_instance)(const char* params_json /* nullable */, CLogCallback log_callback, void* log_user_data);
    CResult (*process_packet)(CPluginHandle handle, const char* input_pin, const CPacket* packet,
                              COutputCallback output_callback, void* callback_data,
                              CTelemetryCallback telemetry_callback, void* telemetry_user_data);
    CResult (*update_params)(CPluginHandle handle, const char* params_json /* nullable */);
    CResult (*flush)(CPluginHandle handle, COutputCallback output_callback, void* callback_data,
                     CTelemetryCallback telemetry_callback, void* telemetry_user_data);
    void (*destroy_instance)(CPluginHandle handle);
} CNativePluginAPI;

/* the one symbol a plugin exports (types.rs:264; plugin-native lib.rs:69-87) */
const CNativePluginAPI* streamkit_native_plugin_api(void);

#ifdef __cplusplus
}
#endif
#endif
